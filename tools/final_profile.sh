# Round-5 final evidence, run on the GPU box in three gpurun calls (each bounded):   bash tools/final_profile.sh 1|2|3
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
case "$1" in
1)  # fp32 headline: kernel-trace stats + the PMC passes (hash-tied), the driver's bench line, the other precisions' stats
    bash tools/profile_gpu.sh r5z > gpurun_out/r5z_profile.log 2>&1
    for p in bf16_fc fp32_split fp32_f16x2; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5z_stats_$p -o r5z_$p -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing --precision $p > gpurun_out/r5z_stats_$p.log 2>&1; done
    python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5z_bench.json 2> gpurun_out/r5z_bench.err
    tail -c 400 gpurun_out/r5z_bench.json ;;
2)  # the driver's bench line once more (profiles/pmc_latest.json now carries this build's hash: roofline.traffic is not stale), the other
    # precisions' bench lines, fp32_f16x2's counters, latency tools, the GPU suite
    python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5z_bench.json 2> gpurun_out/r5z_bench.err
    python bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16_fc --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_bf16.json 2>> gpurun_out/r5z_bench.err
    python bench.py --gpus 1 --steps 20 --warmup 5 --precision fp32_f16x2 --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_f16x2.json 2>> gpurun_out/r5z_bench.err
    python bench.py --gpus 1 --steps 20 --warmup 5 --precision fp32_split --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_fp32_split.json 2>> gpurun_out/r5z_bench.err
    bash tools/profile_f16x2.sh r5z_f16x2 > /dev/null 2>&1
    python tools/latency_small_batch.py > gpurun_out/r5z_latency.txt 2>&1
    env -u DCE_LAT_TRACE python tools/latency_mode.py > gpurun_out/r5z_latency_mode.json 2> gpurun_out/r5z_latency_mode.err
    timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r5z_gpu_tests_full.txt 2>&1; tail -3 gpurun_out/r5z_gpu_tests_full.txt > gpurun_out/r5z_gpu_tests.txt
    cat gpurun_out/r5z_gpu_tests.txt ;;
3)  # precision audit (all four evaluations), randomised sweeps, soak
    timeout 2400 python tools/precision_audit.py > gpurun_out/r5_precision_audit.json 2> gpurun_out/r5_precision_audit.err
    PRECISION=fp32_f16x2 TRIALS=60 timeout 600 python tools/fuzz_parity_x3.py > gpurun_out/r5z_fuzz_f16x2.json 2> gpurun_out/r5z_fuzz.err
    TRIALS=40 timeout 600 python tools/fuzz_parity_x3.py > gpurun_out/r5z_fuzz_fp32_split.json 2>> gpurun_out/r5z_fuzz.err
    timeout 600 python tools/soak_x3.py fp32_f16x2 fp32_split > gpurun_out/r5z_soak.json 2> gpurun_out/r5z_soak.err
    tail -5 gpurun_out/r5_precision_audit.err; cat gpurun_out/r5z_fuzz_f16x2.json gpurun_out/r5z_fuzz_fp32_split.json gpurun_out/r5z_soak.json ;;
esac

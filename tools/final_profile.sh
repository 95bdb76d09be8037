# Round-6 final evidence, run on the GPU box in bounded gpurun calls:   bash tools/final_profile.sh 1|2|3
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
T=r6z
case "$1" in
1)  # fp32 headline: kernel-trace stats + the PMC passes (hash-tied), the other precisions' stats, the driver's bench line
    bash tools/profile_gpu.sh $T > gpurun_out/${T}_profile.log 2>&1
    for p in bf16_fc fp32_f16x2; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats_$p -o ${T}_$p -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing --precision $p > gpurun_out/${T}_stats_$p.log 2>&1; done
    bash tools/profile_f16x2.sh ${T}_f16x2 > /dev/null 2>&1
    # the latency mode's micro-batch kernel under the profiler: 30-window calls
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${T}_stats_latmb -o ${T}_latmb -- python tools/dev_latmb.py 8,30 > gpurun_out/${T}_stats_latmb.log 2>&1
    python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
    tail -c 300 gpurun_out/${T}_bench.json ;;
2)  # the driver's bench line once more (profiles/pmc_latest.json now carries this build's hash: roofline.traffic is not stale), the other
    # precisions' bench lines, latency tools, the GPU suite, smoke
    python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err
    python bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16_fc --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_bf16.json 2>> gpurun_out/${T}_bench.err
    python bench.py --gpus 1 --steps 20 --warmup 5 --precision fp32_f16x2 --no-cpu-baseline --no-extras > gpurun_out/${T}_bench_f16x2.json 2>> gpurun_out/${T}_bench.err
    python tools/latency_small_batch.py > gpurun_out/${T}_latency.txt 2>&1
    DCE_LAT_TRACE=1 python tools/dev_latmb.py 2,8,16,17,30,32 2>&1 | grep -v amdgpu.ids > gpurun_out/${T}_latmb.txt
    timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/${T}_gpu_tests_full.txt 2>&1; tail -3 gpurun_out/${T}_gpu_tests_full.txt > gpurun_out/${T}_gpu_tests.txt
    python __graft_entry__.py --smoke 2>&1 | tail -1 >> gpurun_out/${T}_gpu_tests.txt
    cat gpurun_out/${T}_gpu_tests.txt ;;
3)  # every one of 1e6 windows against the oracle on the final sources (fp32; the two 16-bit precisions on the AR(1) 200k set), soak of the latency mode
    timeout 1500 python tools/check_full_parity.py > gpurun_out/${T}_full_parity_1e6.json 2> gpurun_out/${T}_full_parity.err
    KIND=ar1 N_WINDOWS=200000 PRECISION=fp32_f16x2 timeout 900 python tools/check_full_parity.py > gpurun_out/${T}_full_parity_ar1_200k_f16x2.json 2>> gpurun_out/${T}_full_parity.err
    KIND=ar1 N_WINDOWS=200000 timeout 900 python tools/check_full_parity.py > gpurun_out/${T}_full_parity_ar1_200k.json 2>> gpurun_out/${T}_full_parity.err
    cat gpurun_out/${T}_full_parity_1e6.json gpurun_out/${T}_full_parity_ar1_200k_f16x2.json gpurun_out/${T}_full_parity_ar1_200k.json | cut -c1-700 ;;
esac

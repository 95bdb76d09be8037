mkdir -p gpurun_out; bash tools/profile_gpu.sh r5z > gpurun_out/r5z_profile.log 2>&1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for p in bf16_fc fp32_split; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r5z_stats_$p -o r5z_$p -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing --precision $p > gpurun_out/r5z_stats_$p.log 2>&1; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5z_bench.json 2> gpurun_out/r5z_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16_fc --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_bf16.json 2>> gpurun_out/r5z_bench.err
python tools/latency_small_batch.py > gpurun_out/r5z_latency.txt 2>&1
env -u DCE_LAT_TRACE python tools/latency_mode.py > gpurun_out/r5z_latency_mode.json 2> gpurun_out/r5z_latency_mode.err
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r5z_gpu_tests_full.txt 2>&1; tail -3 gpurun_out/r5z_gpu_tests_full.txt > gpurun_out/r5z_gpu_tests.txt
timeout 1500 python tools/precision_audit.py > gpurun_out/r5_precision_audit.json 2> gpurun_out/r5_precision_audit.err
tail -c 600 gpurun_out/r5z_bench.json; cat gpurun_out/r5z_gpu_tests.txt; tail -5 gpurun_out/r5_precision_audit.err

mkdir -p gpurun_out; bash tools/profile_gpu.sh r4z > gpurun_out/r4z_profile.log 2>&1
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for p in bf16_fc fp32_split; do rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4z_stats_$p -o r4z_$p -- python bench.py --steps 300 --warmup 50 --no-cpu-baseline --no-extras --no-kernel-timing --precision $p > gpurun_out/r4z_stats_$p.log 2>&1; done
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r4z_bench.json 2> gpurun_out/r4z_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16_fc --no-cpu-baseline --no-extras > gpurun_out/r4z_bench_bf16.json 2>> gpurun_out/r4z_bench.err
python tools/latency_small_batch.py > gpurun_out/r4z_latency.txt 2>&1
python tools/online_latency.py > gpurun_out/r4z_online.txt 2>&1
tail -c 1500 gpurun_out/r4z_bench.json; ls gpurun_out | grep r4z | head -40

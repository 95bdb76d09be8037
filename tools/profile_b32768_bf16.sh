cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4m
B="python bench.py --steps 20 --warmup 3 --settle-s 0.2 --no-cpu-baseline --no-extras --no-kernel-timing --precision bf16_fc --batch 32768"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4m/stats -o b32k -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-kernel-timing --precision bf16_fc --batch 32768 > gpurun_out/r4m/stats.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r4m/pmc -o b32k -- $B > gpurun_out/r4m/pmc.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --precision bf16_fc --batch 32768 > gpurun_out/r4m/bench_bf16_b32768.json 2>/dev/null
python tools/pmc_summary.py gpurun_out/r4m/pmc/b32k_counter_collection.csv > gpurun_out/r4m/pmc_summary.txt 2>&1
head -6 gpurun_out/r4m/stats/b32k_kernel_stats.csv | cut -c1-200; cat gpurun_out/r4m/pmc_summary.txt | head -20; cut -c1-400 gpurun_out/r4m/bench_bf16_b32768.json

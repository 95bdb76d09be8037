python -m pytest tests/test_round3_gpu.py -m gpu -q -k "bf16_fc_vs_independent" 2>&1 | tail -3
for t in "" "x3_bf16_terms=3" "bf16_conv_h2=1"; do DCE_TUNE=$t python bench.py --precision bf16_fc --steps 300 --warmup 50 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('bf16_fc $t', round(d['value']/1e6,3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done

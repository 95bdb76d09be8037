mkdir -p gpurun_out
python -m pytest tests/test_f16x2_gpu.py tests/test_x3_gpu.py -m gpu -q -x 2>&1 | tail -8
for t in "" "h2_fc3=0"; do DCE_TUNE=$t python bench.py --precision fp32_f16x2 --steps 300 --warmup 50 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('f16x2 $t', round(d['value']/1e6,3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done
python bench.py --precision fp32_split --steps 300 --warmup 50 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('fp32_split', round(d['value']/1e6,3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"

export DCE_LIB=$PWD/deep_contact_estimator_amd/libdce_experiments.so
for t in "h2_fc3=0" "h2_fc3=0,h2_ksplit=1" "h2_ksplit=1"; do DCE_TUNE=$t python bench.py --precision fp32_f16x2 --steps 300 --warmup 50 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('f16x2 $t', round(d['value']/1e6,3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done
python -m pytest tests/test_f16x2_gpu.py -m gpu -q -k "dealt_out" 2>&1 | tail -3

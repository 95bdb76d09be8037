python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5z_bench.json 2> gpurun_out/r5z_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --precision bf16_fc --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_bf16.json 2>> gpurun_out/r5z_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --precision fp32_f16x2 --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_f16x2.json 2>> gpurun_out/r5z_bench.err
python bench.py --gpus 1 --steps 20 --warmup 5 --precision fp32_split --no-cpu-baseline --no-extras > gpurun_out/r5z_bench_fp32_split.json 2>> gpurun_out/r5z_bench.err
python -c "
import json
d=json.load(open('gpurun_out/r5z_bench.json'));print(d['value'], d['roofline']['frac'], d['roofline']['traffic_stale'], {k: round(v['windows_per_s']/1e6,2) for k,v in d['precisions'].items()})
b=d['extra']['bf16_fc'];print({k: round(b[k]['windows_per_s']/1e6,2) for k in ('two_term_bf16_conv_stack','three_term_conv_stack')}, b['conv_stack'][:60])
for f in ('bf16','f16x2','fp32_split'):
    e=json.load(open('gpurun_out/r5z_bench_%s.json'%f)); print(f, round(e['value']/1e6,3), round(e['roofline']['frac'],3), e['roofline'].get('traffic_stale'), {k:round(v['avg_ms']*1e3,1) for k,v in e['kernels'].items()})
"
python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3
python __graft_entry__.py --smoke 2>&1 | tail -1

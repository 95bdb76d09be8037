for r in 1 2; do for g in 1 0; do
DCE_TUNE=split_guard=$g python bench.py --precision fp32_split --no-extras --no-cpu-baseline --no-kernel-timing --steps 300 --warmup 20 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('guard=$g round=$r', round(j['value']/1e6,3), 'M  ', round(j['ms_per_step']*1e3,1), 'us')"
done; done

for t in "" "bf16_fc3_ksplit=1"; do DCE_TUNE=$t python bench.py --precision bf16_fc --steps 300 --warmup 50 --no-cpu-baseline --no-extras > gpurun_out/ab.json 2>/dev/null; python -c "
import json;d=json.load(open('gpurun_out/ab.json'));print('bf16_fc $t', round(d['value']/1e6,3), {k:round(v['avg_ms']*1e3,1) for k,v in d['kernels'].items()})"; done
python - <<'PY'
import numpy as np
from deep_contact_estimator_amd import contact_cnn, synth
sd = synth.make_state_dict(1, "uniform")
a = contact_cnn(device=0, max_batch=8192, precision="bf16_fc"); a.load_state_dict(sd).eval()
b = contact_cnn(device=0, max_batch=8192, precision="bf16_fc", tune={"bf16_fc3_ksplit": 1}); b.load_state_dict(sd).eval()
for n in (3072, 4096, 4100, 8192):
    x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
    x[n - 1, 3, 3] = np.nan
    ra, rb = a.predict(x), b.predict(x)
    d = np.nanmax(np.abs(ra["logits"] - rb["logits"])); s = np.nanmax(np.abs(ra["logits"]))
    print(n, b.last_plan(), "max|d| / scale", d / s, "pred diff", int((ra["pred"] != rb["pred"]).sum()), "nan row ok", bool(np.isnan(rb["logits"][n - 1]).all()))
PY

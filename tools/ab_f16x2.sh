mkdir -p gpurun_out
python -m pytest tests/test_f16x2_gpu.py -m gpu -q -x 2>&1 | tail -4
python - <<'PY'
import time, torch, numpy as np
from deep_contact_estimator_amd import contact_cnn, synth
sd = synth.make_state_dict(1, "uniform")
g = torch.Generator(device="cuda").manual_seed(3)
seq = torch.randn((1_000_000 + 149, 54), generator=g, device="cuda", dtype=torch.float32)
for tune in (None, {"h2_fc3": 0}):
    for prec in ("fp32_f16x2",):
        m = contact_cnn(device=0, max_batch=32768, precision=prec, tune=tune); m.load_state_dict(sd).eval()
        m.infer_sequence(seq[:32768 + 149]); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); m.infer_sequence(seq); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        print(prec, tune, "streaming 1e6:", round(1e6 / sorted(ts)[1] / 1e6, 3), "M windows/s", m.last_plan())
        m.close()
PY

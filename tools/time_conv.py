#!/usr/bin/env python3
"""Event-timed kernels of the bench step for one library build (DCE_LIB) and precision; prints one line.
    python tools/time_conv.py <precision> [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16_fc"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
B = int(os.environ.get("TIME_B", "4096"))
m = contact_cnn(device=0, max_batch=B, precision=prec); m.load_state_dict(synth.make_state_dict(1)).eval()
x = torch.randn((B, 150, 54), device="cuda")
for _ in range(30): m.predict_packed(x)
m.sync(); m.profile(1); m.profile_read()
t0 = time.perf_counter()
for _ in range(steps): m.predict_packed(x)
m.sync()
dt = (time.perf_counter() - t0) / steps
pr = m.profile_read()
print(f"{os.environ.get('DCE_LIB', 'default'):50s} {prec:10s} B={B} step {dt * 1e6:7.1f} us  {B / dt / 1e6:6.3f} M/s  " +
      "  ".join(f"{k} {v['ms'] / max(v['launches'], 1) * 1e3:6.1f}" for k, v in pr.items()) + "  plan " + " ".join(m.last_plan()), flush=True)

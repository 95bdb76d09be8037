#!/usr/bin/env python3
"""Latency / throughput at the reference's shipped batch sizes (1: inference_one_seq_params.yaml:10,
30: test_params.yaml:9) through the per-batch API (model.predict on a device tensor)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
m = contact_cnn(device=0, max_batch=4096, precision=os.environ.get("DCE_LAT_PRECISION", "fp32")); m.load_state_dict(synth.make_state_dict(1))
seq = torch.from_numpy(synth.make_sequence(4096 + 149, 2).astype(np.float32)).cuda()
x = m.zscore_windows(seq)
res = {}
sizes = [int(v) for v in os.environ.get("DCE_LAT_SIZES", "1,2,30,64,128,256,512,1024,2048,3000,4096").split(",")]
for B in sizes:
    xb = x[:B].contiguous()
    for _ in range(20): m.predict(xb)
    torch.cuda.synchronize()
    n = 300
    t0 = time.perf_counter()
    for _ in range(n): m.predict(xb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    m.profile(1); m.profile_read(True)
    for _ in range(50): m.predict(xb)
    torch.cuda.synchronize(); p = m.profile_read(True); m.profile(0)
    res[B] = {"us_per_call": dt * 1e6, "windows_per_s": B / dt, "kernel_us": {k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in p.items() if v["launches"]}}
    res[B]["plan"] = " ".join(m.last_plan())
    print(B, json.dumps(res[B]))

#!/usr/bin/env python3
"""BASELINE.json configs[2]: full-sequence streaming inference on 1e6 synthetic windows.
Reports (a) HBM-resident rate (sequence on device, outputs on device: what bench.py's metric
would be on this workload) and (b) the PCIe-inclusive rate when the boundary is handed host
buffers (numpy (T,54) in, logits/pred/contacts out), which is never bench.py's `value`."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth

N = int(os.environ.get("N_WINDOWS", 1_000_000))
prec = os.environ.get("PRECISION", "fp32")
mb = int(os.environ.get("MAX_BATCH", 32768))
m = contact_cnn(device=0, max_batch=mb, precision=prec)
m.load_state_dict(synth.make_state_dict(1)).eval()
g = torch.Generator(device="cuda").manual_seed(3)
seq = torch.randn((N + 149, 54), generator=g, device="cuda", dtype=torch.float32)
m.infer_sequence(seq[: 4096 + 149])
torch.cuda.synchronize()
reps = 3
t0 = time.perf_counter()
for _ in range(reps):
    out = m.infer_sequence(seq)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
host = seq.cpu().numpy()
m.infer_sequence(host[: 4096 + 149])
dths = []
for _ in range(3):                       # first call also grows the ctx's staging buffers
    t0 = time.perf_counter()
    out_h = m.infer_sequence(host)
    dths.append(time.perf_counter() - t0)
dth = min(dths)
assert np.array_equal(out_h["contacts"], out["contacts"].cpu().numpy())
print(json.dumps({
    "workload": f"configs[2]: infer_sequence over {N} windows (T={N + 149}), max_batch {mb}, {prec}",
    "hbm_resident_windows_per_s": N / dt, "hbm_resident_ms": dt * 1e3,
    "pcie_inclusive_windows_per_s": N / dth, "pcie_inclusive_ms": dth * 1e3,
    "pcie_inclusive_ms_each_call": [round(d * 1e3, 1) for d in dths],
    "bytes_h2d": int(host.nbytes), "bytes_d2h": int(N * (64 + 4 + 4)),
    "classes_seen": int(len(np.unique(out_h["pred"]))),
}))

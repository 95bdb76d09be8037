#!/usr/bin/env python3
"""Per-call time of model.predict on n device-resident windows, fp32 against fp32_f16x2, around the mode's thresholds (128: conv_h2_f32 + fp32 FC
kernels; 2817: the two-term FC kernels).  Run on the GPU box."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
sd = synth.make_state_dict(1, "uniform")
ms = {p: contact_cnn(device=0, max_batch=8192, precision=p) for p in ("fp32", "fp32_f16x2")}
for m in ms.values(): m.load_state_dict(sd).eval()
x = torch.randn((8192, 150, 54), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
out = {}
for n in (64, 127, 128, 192, 256, 384, 512, 768, 1024, 1536, 2048, 2816, 2817, 3072, 4096, 8192):
    row = {}
    for p, m in ms.items():
        for _ in range(20): m.predict(x[:n])
        torch.cuda.synchronize()
        reps = 300 if n <= 1024 else 100
        t0 = time.perf_counter()
        for _ in range(reps): m.predict(x[:n])
        torch.cuda.synchronize()
        row[p] = round((time.perf_counter() - t0) / reps * 1e6, 1)
        row[p + "_plan"] = m.last_plan()[:2]
    row["ratio"] = round(row["fp32"] / row["fp32_f16x2"], 2)
    out[n] = row
    print(n, row, flush=True)
print(json.dumps(out))

#!/usr/bin/env python3
"""Round 6: the record on which DCE_FP32_SPLIT leaves the product library.  Per-call time of model.predict on n device-resident windows
in the two precisions that hold the fp32 TOLERANCE on the 16-bit matrix pipes -- fp32_split (three bf16 terms, six MFMAs per product, range-guarded) and
fp32_f16x2 (two fp16 terms with per-window scales, three MFMAs per product) -- from the first size where either leaves the fp32 kernels (128) up.
The accuracy side is profiles/r5_precision_audit.json (err / bound against the fp64 oracle on every audit set).  Run on the GPU box, product library of round 5 /
experiments library of round 6."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
sd = synth.make_state_dict(1, "uniform")
ms = {p: contact_cnn(device=0, max_batch=32768, precision=p) for p in ("fp32_split", "fp32_f16x2")}
for m in ms.values(): m.load_state_dict(sd).eval()
x = torch.randn((32768, 150, 54), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
out = {}
for n in (128, 192, 256, 512, 1024, 1280, 1281, 2048, 2816, 2817, 3072, 4096, 8192, 12288, 12289, 16384, 32768):
    row = {}
    for p, m in ms.items():
        for _ in range(10): m.predict(x[:n])
        torch.cuda.synchronize()
        reps = 200 if n <= 4096 else 40
        t0 = time.perf_counter()
        for _ in range(reps): m.predict(x[:n])
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        row[p] = {"us": round(dt * 1e6, 1), "M_windows_per_s": round(n / dt / 1e6, 3), "plan": m.last_plan()[:3]}
    row["f16x2_over_split"] = round(row["fp32_f16x2"]["M_windows_per_s"] / row["fp32_split"]["M_windows_per_s"], 3)
    out[n] = row
    print(n, json.dumps(row), flush=True)
print(json.dumps({"f16x2_at_least_as_fast_at_every_size": all(r["f16x2_over_split"] >= 0.995 for r in out.values()), "sizes": out}))

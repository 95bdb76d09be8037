#!/usr/bin/env python3
"""Round 5's one experiment on the bf16 fc.0 GEMM: the 256 x 128 tile with 32-k K-tiles (3 x 24 KB of LDS, <= 128
registers) so that TWO workgroups share a CU, against the shipped 64-k K-tiles with one workgroup per CU -- at 8192 and 32768
windows per launch (the tile count it needs: >= 512).  Same MFMA sequence per output: the results must be the same bits."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth

sd = synth.make_state_dict(1, "uniform")
res = {}
for n in (8192, 32768):
    x = torch.randn((n, 150, 54), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
    outs, r = {}, {}
    for rounds in range(2):                                   # interleaved rounds: the board's clock drifts
        for tag, tune in (("k64_one_per_cu", {"bf16_k32": 0}), ("k32_two_per_cu", {"bf16_k32": 1})):
            m = contact_cnn(device=0, max_batch=n, precision="bf16_fc", tune=tune); m.load_state_dict(sd).eval()
            for _ in range(30): m.predict(x)
            torch.cuda.synchronize()
            m.profile(1)
            t0 = time.perf_counter()
            for _ in range(100): o = m.predict(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100
            p = m.profile_read()
            outs[tag] = o["logits"].clone()
            r.setdefault(tag, []).append({"fc1_gemm_us": p["fc1_gemm"]["ms"] / p["fc1_gemm"]["launches"] * 1e3, "step_us_with_events": dt * 1e6,
                                          "frac_of_2.5PF": n * 19398656 / (p["fc1_gemm"]["ms"] / p["fc1_gemm"]["launches"] * 1e-3) / 2.5e15, "plan": " ".join(m.last_plan())})
            m.close()
    r["bit_identical"] = bool(torch.equal(outs["k64_one_per_cu"], outs["k32_two_per_cu"]))
    res[str(n)] = r
print(json.dumps(res, indent=1))

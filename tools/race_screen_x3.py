#!/usr/bin/env python3
"""Race screen for the kernels of the fp32_split precision (conv_x3.hip: in-place LDS write-backs between barriers;
fc_gemm_x3.hip: LDS-DMA into a two-buffer ring ordered against fragment reads by counted waits + barriers): the arithmetic is
deterministic, so every repeated run must return the bits of the first -- also in a second, fresh context, and while a second
stream keeps the memory system busy (uneven load shifts LDS-DMA landing times)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deep_contact_estimator_amd import contact_cnn, synth

sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "3072,4096,5003,12288,32768").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 150
sd = synth.make_state_dict(1, "uniform")
PREC = os.environ.get("DCE_RACE_PRECISION", "fp32_split")        # bf16_fc: the two-term conv stack (three workgroups per CU) + the bf16 GEMMs
noise_src = torch.empty(64 << 20, dtype=torch.uint8, device="cuda")
noise_dst = torch.empty_like(noise_src)
side = torch.cuda.Stream()
out, plans = {}, {}
t0 = time.time()
for n in sizes:
    x = torch.randn((n, 150, 54), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
    a = contact_cnn(device=0, max_batch=n, precision=PREC); a.load_state_dict(sd)
    ref = a.predict(x)["logits"].clone()
    plans[n] = a.last_plan()
    b = contact_cnn(device=0, max_batch=n, precision=PREC); b.load_state_dict(sd)
    bad = int(not torch.equal(b.predict(x)["logits"], ref))
    for r in range(reps):
        if r % 2:                                   # every other run under a concurrent copy stream
            with torch.cuda.stream(side):
                for _ in range(4):
                    noise_dst.copy_(noise_src, non_blocking=True)
        m = a if r % 3 else b
        bad += int(not torch.equal(m.predict(x)["logits"], ref))
    torch.cuda.synchronize()
    a.close(); b.close()
    out[n] = bad
print(json.dumps({"precision": PREC, "reps": reps, "mismatching_runs": out, "kernels": plans, "seconds": round(time.time() - t0, 1)}))

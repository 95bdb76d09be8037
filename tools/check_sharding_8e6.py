#!/usr/bin/env python3
"""BASELINE.json configs[3] at full size on ONE GPU: 8e6 windows (T = 8,000,149 rows, 1.73 GB)
through dce_infer_sequence in one call, and again as the 8 halo-sharded ranges
distributed.shard_rows gives the 8 ranks of a node (run one after the other here).  The
concatenated shard results must equal the single-call results bit for bit -- the sharding adds no
arithmetic -- and both must satisfy the size-independent properties (argmax of the returned logits,
MSB-first contact bits, determinism)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
from deep_contact_estimator_amd.distributed import shard_rows

N = int(os.environ.get("N_WINDOWS", 8_000_000))
G = 8
T = N + 149
m = contact_cnn(device=0, max_batch=32768)
m.load_state_dict(synth.make_state_dict(1, "uniform")).eval()
g = torch.Generator(device="cuda").manual_seed(3)
seq = torch.randn((T, 54), generator=g, device="cuda", dtype=torch.float32)
m.infer_sequence(seq[:4096 + 149]); torch.cuda.synchronize()
t0 = time.perf_counter()
full = m.infer_sequence(seq)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
again = m.infer_sequence(seq)
det = all(torch.equal(full[k], again[k]) for k in full)
del again
parts = []
for r in range(G):
    r0, r1, w0, w1 = shard_rows(T, r, G)
    parts.append(m.infer_sequence(seq[r0:r1]))
    assert parts[-1]["pred"].shape[0] == w1 - w0
same = all(torch.equal(torch.cat([p[k] for p in parts], 0), full[k]) for k in full)
lg, pr, ct = full["logits"], full["pred"], full["contacts"]
am = torch.argmax(lg, 1).to(pr.dtype)
bits = ((pr[:, None] >> torch.tensor([3, 2, 1, 0], device=pr.device)) & 1).to(torch.uint8)
print(json.dumps({
    "workload": f"configs[3] on one GPU: {N} windows, one call vs {G} halo shards (149-row halo each)",
    "single_call_ms": dt * 1e3, "windows_per_s": N / dt,
    "shards_equal_single_call_bitwise": bool(same), "deterministic": bool(det),
    "pred_is_argmax_of_logits": bool(torch.equal(am, pr)), "contacts_are_bits_of_pred": bool(torch.equal(bits, ct)),
    "finite_logits": bool(torch.isfinite(lg).all()), "classes_seen": int(torch.unique(pr).numel()),
}))

#!/usr/bin/env python3
"""BASELINE.json configs[2], literally: contacts of the 1e6-window streaming run compared with the
CPU oracle on EVERY window (OpenMP over the host cores).  Reports flips and their fp32 margins."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
from oracle import oracle as orc

N = int(os.environ.get("N_WINDOWS", 1_000_000))
sd = synth.make_state_dict(1)
KIND = os.environ.get("KIND", "normal")
if KIND == "normal":
    seq = np.random.default_rng(3).standard_normal((N + 149, 54)).astype(np.float32)
else:                                   # AR(1) + per-channel offset/scale: stresses the z-score
    seq = synth.make_sequence(N + 149, 5, "ar1").astype(np.float32)
PRECISION = os.environ.get("PRECISION", "fp32")     # fp32 | fp32_split (fc.0 on three-term bf16 operands): same contract
m = contact_cnn(device=0, precision=PRECISION)
m.load_state_dict(sd)
t0 = time.time(); out = m.infer_sequence(seq); tg = time.time() - t0
t0 = time.time()
if PRECISION == "bf16_fc":                  # the mode's own CPU restatement, chunk by chunk through the materialised-window entry
    o16, parts = orc.Oracle(sd, bf16_fc=True), []
    for b0 in range(0, N, 20000):
        b1 = min(N, b0 + 20000)
        parts.append(o16.forward_windows(orc.zscore_windows(seq[b0:b1 + 149])))
    ref = {k: np.concatenate([p[k] for p in parts]) for k in ("logits", "pred", "contacts")}
else:
    ref = orc.Oracle(sd).infer_sequence(seq)
tc = time.time() - t0
flips = np.nonzero(out["pred"] != ref["pred"])[0]
srt = np.sort(ref["logits"], axis=1); margin = srt[:, -1] - srt[:, -2]
err = np.abs(out["logits"] - ref["logits"])
bound = 1e-5 * np.abs(ref["logits"]).max() + 1e-4 * np.abs(ref["logits"])
print(json.dumps({
    "windows": N, "kind": KIND, "precision": PRECISION, "plan_of_last_chunk": m.last_plan(), "gpu_s_incl_pcie": tg, "oracle_s": tc, "oracle_threads": os.cpu_count(),
    "argmax_flips": int(flips.size), "flip_margins": margin[flips][:20].tolist(),
    "contacts_equal_rows": int((out["contacts"] == ref["contacts"]).all(axis=1).sum()),
    "max_abs_logit_err": float(err.max()), "max_err_over_bound": float((err / bound).max()),
    "min_margin": float(margin.min()), "windows_with_margin_below_1e-4": int((margin < 1e-4).sum()),
    "classes_seen": int(np.unique(ref["pred"]).size),
    # (bf16_fc: the contract is a band on the logits -- rounding-boundary flips of bf16 features / h1 entries -- and argmax equality above a 1e-2 margin)
    "max_abs_logit_err_over_largest_logit": float(err.max() / np.abs(ref["logits"]).max()),
    "argmax_flips_above_1e-2_margin": int((margin[flips] > 1e-2 * np.abs(ref["logits"]).max()).sum())}))

#!/usr/bin/env python3
"""Soak of the latency mode (run on the GPU box): ~200k online pushes through the resident service kernel, every estimate compared
with the batch path's result for the same window (fp32 contract; argmax outside the noise margin) -- across resets, idle exits of the
service (latency_idle_ms) and interleaved one-window calls -- then 20,000 one-window calls (bit-stable), and create / destroy cycles."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth

sd = synth.make_state_dict(1, "uniform")
seq = synth.make_sequence(150 + 16000, 5).astype(np.float32)
b = contact_cnn(device=0, max_batch=4096); b.load_state_dict(sd).eval()
ref = b.infer_sequence(seq)
scale = np.abs(ref["logits"]).max()
srt = np.sort(ref["logits"], axis=1); safe = (srt[:, -1] - srt[:, -2]) > 1e-3 * scale
m = contact_cnn(device=0, max_batch=64, tune={"latency": 1, "latency_idle_ms": 20}); m.load_state_dict(sd).eval()
free0 = torch.cuda.mem_get_info()[0]
worst, n_ok, flips, restarts = 0.0, 0, 0, 0
t0 = time.time()
for rep in range(12):
    m.online_reset()
    for t in range(len(seq)):
        if t % 5000 == 4999:
            time.sleep(0.05); restarts += 1                      # the service leaves by itself
        if t % 7001 == 7000:
            m.predict(seq[:150][None]); restarts += 1             # ... or for another call
        r = m.online_push(seq[t])
        if r is None: continue
        j = t - 149
        e = np.abs(r[0] - ref["logits"][j]) / (1e-5 * scale + 1e-4 * np.abs(ref["logits"][j]))
        worst = max(worst, float(e.max())); n_ok += 1
        if r[1] != ref["pred"][j]:
            flips += 1
            if safe[j]: raise SystemExit(f"argmax differs above the margin: rep {rep} t {t}")
        if e.max() > 1.0: raise SystemExit(f"outside the contract: rep {rep} t {t} err/bound {e.max():.3f}")
dt = time.time() - t0
x = m.zscore_windows(torch.from_numpy(seq[:150]).cuda())
first = m.predict(x)["logits"].clone()
for i in range(20000): out = m.predict(x)
torch.cuda.synchronize()
same = bool(torch.equal(out["logits"], first))
for i in range(100):
    c = contact_cnn(device=0, max_batch=16, tune={"latency": 1}); c.load_state_dict(sd)
    [c.online_push(seq[t]) for t in range(155)]; c.predict(seq[:150][None]); c.close()
m.close(); b.close(); torch.cuda.empty_cache()
print(json.dumps({"pushes_checked": n_ok, "worst_err_over_bound_vs_batch_path": worst, "sub_margin_argmax_differences": flips, "service_restarts": restarts,
                  "us_per_push_incl_checks": dt / (12 * len(seq)) * 1e6, "one_window_calls_20000_bit_stable": same, "ctx_cycles": 100,
                  "device_memory_delta_MB": (free0 - torch.cuda.mem_get_info()[0]) / 1e6}))

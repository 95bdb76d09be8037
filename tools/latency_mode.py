#!/usr/bin/env python3
"""Latency mode (option latency=1, csrc/latency.hip) against the batch path at the reference's shipped batch_size 1: device time per
one-window predict() (200 stream-ordered calls on a device-resident window, as bench.py's extra.small_batches measures) and host
time per online push (sample in -> estimate out)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth

sd = synth.make_state_dict(1, "uniform")
seq = synth.make_sequence(150 + 2400, seed=2).astype(np.float32)
res = {}
os.environ["DCE_LAT_TRACE"] = "1"
import ctypes as C
def trace(m):
    t = (C.c_uint64 * 16)()
    if m._lib.dce_debug_latency_trace(m._ctx, t) != 0: return None
    t = list(t); base = min(v for v in t if v)
    return {f"t{i}_us": (v - base) / 100.0 for i, v in enumerate(t) if v}
for tag, tune in (("batch_path", None), ("latency_mode", {"latency": 1})):
    m = contact_cnn(device=0, max_batch=64, tune=tune); m.load_state_dict(sd).eval()
    x = m.zscore_windows(torch.from_numpy(seq[:150 + 63]).cuda())
    r = {}
    for b in (1, 2):
        xb = x[:b].contiguous()
        for _ in range(50): m.predict(xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(500): m.predict(xb)
        torch.cuda.synchronize()
        r[f"predict_{b}_us"] = (time.perf_counter() - t0) / 500 * 1e6
        r[f"plan_{b}"] = " ".join(m.last_plan())
        if b == 1 and tune: r["trace_one_shot"] = trace(m)
    xh = x[:1].cpu().numpy()
    for _ in range(20): m.predict(xh)
    t0 = time.perf_counter()
    for _ in range(300): m.predict(xh)
    r["predict_1_numpy_in_out_us"] = (time.perf_counter() - t0) / 300 * 1e6
    m.online_reset()
    for t in range(150 + 200): m.online_push(seq[t])
    lat = []
    for t in range(350, seq.shape[0]):
        t0 = time.perf_counter(); m.online_push(seq[t]); lat.append(time.perf_counter() - t0)
    lat = np.array(lat) * 1e6
    if tune: r["trace_push"] = trace(m)
    r["online_push_us"] = {"mean": float(lat.mean()), "p50": float(np.percentile(lat, 50)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max()), "pushes": int(lat.size)}
    m.close()
    res[tag] = r
print(json.dumps(res, indent=1))

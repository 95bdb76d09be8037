#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of the conv-stack kernel (needs the trace variant:
build_variant('trace', ['-DDCE_TRACE=1']) and DCE_LIB pointing at it)."""
import ctypes as C
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import contact_cnn, synth, _lib

B = 4096
m = contact_cnn(device=0, max_batch=B)
m.load_state_dict(synth.make_state_dict(1)).eval()
x = np.random.default_rng(0).standard_normal((B, 150, 54), dtype=np.float32)
import torch
xd = torch.from_numpy(x).cuda()
for _ in range(3):
    m.predict(xd)
torch.cuda.synchronize()
lib = _lib.load()
nb = B // 2
buf = np.zeros((nb, 16), np.uint64)
reader = lib.dce_debug_trace_read if 'conv_direct=1' in os.environ.get('DCE_TUNE', '') else lib.dce_debug_trace_read_wino      # (the direct-form stack: DCE_TUNE=conv_direct=1, experiments build)
rc = reader(buf.ctypes.data_as(C.c_void_p), nb)
assert rc == 0
t = buf[:, :10].astype(np.int64)
hw = buf[:, 10].astype(np.int64)
t0 = t[:, 0].min()
names = ["prologue", "conv1", "store1", "conv2", "store2", "conv3", "store3", "conv4", "store4"]
d = np.diff(t, axis=1)
print("phase durations (cycles): mean / p10 / p90")
for i, nme in enumerate(names):
    print(f"  {nme:9s} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 10):9.0f} {np.percentile(d[:, i], 90):9.0f}")
if buf[:, 11].any():
    sub = buf[:, [0, 11, 12, 13, 14, 1]].astype(np.int64)
    ds = np.diff(sub, axis=1)
    for i, nme in enumerate(["bias copy", "loads + NaN scan", "barrier", "LDS transpose stores", "barrier"]):
        print(f"    prologue / {nme:22s} {ds[:, i].mean():9.0f} {np.percentile(ds[:, i], 10):9.0f} {np.percentile(ds[:, i], 90):9.0f}")
tot = t[:, 9] - t[:, 0]
print("block total", tot.mean(), "kernel span", t[:, 9].max() - t0)
mf = d[:, [1, 3, 5, 7]].sum(axis=1)
print("MFMA-phase share of block time:", (mf / tot).mean())
cu = (hw >> 8) & 0xF; se = (hw >> 13) & 0x7; sh = (hw >> 12) & 1; wave = hw & 0xF; simd = (hw >> 4) & 3
print("wave slot histogram:", np.bincount(wave, minlength=8))
# timeline of the blocks that ran on one CU
key = se * 1000 + sh * 100 + cu
xcc = None
for k in np.unique(key)[:2]:
    idx = np.where(key == k)[0]
    idx = idx[np.argsort(t[idx, 0])]
    print(f"CU key {k}: {len(idx)} blocks")
    for b in idx[:12]:
        print("   blk", b, "slot", wave[b], "start", t[b, 0] - t0, "phases", d[b].tolist())

# ---- how the two co-resident workgroups of a CU overlap: time in (both MFMA) / (one MFMA) / (neither)
def intervals(b):
    # MFMA phases are d[1], d[3], d[5], d[7] -> [t1,t2], [t3,t4], [t5,t6], [t7,t8]
    return [(t[b, 1], t[b, 2]), (t[b, 3], t[b, 4]), (t[b, 5], t[b, 6]), (t[b, 7], t[b, 8])]
both = one = none = 0
key8 = key * 8 + (np.arange(nb) % 8)      # HW_ID repeats on every XCD; workgroup b runs on XCD b % 8
for k in np.unique(key8):
    idx = np.where(key8 == k)[0]
    if len(idx) < 4:
        continue
    ev = []
    for b in idx:
        for (a0, a1) in intervals(b):
            ev.append((a0, +1)); ev.append((a1, -1))
    lo, hi = t[idx, 0].min(), t[idx, 9].max()
    # only spans where the kernel is running on this CU (trace holds several launches: split on big gaps)
    ev.sort()
    cur, last = 0, None
    for (tt, dlt) in ev:
        if last is not None and tt - last < 2_000_000:
            span = tt - last
            if cur >= 2: both += span
            elif cur == 1: one += span
            else: none += span
        cur += dlt; last = tt
tot = both + one + none
print("groups", len(np.unique(key8)), "blocks per group", np.bincount(np.unique(key8, return_inverse=True)[1]).tolist()[:8])
print(f"CU time with MFMA phases of: two workgroups {both / tot:.3f}, one {one / tot:.3f}, none {none / tot:.3f}")

# ---- by generation on a CU (2048 workgroups over 512 slots = 4 generations): is the launch-level loss in the first generation
#      (every CU's two workgroups in their prologue at once) or in the tail?
gen_tot, gen_pro, gen_start, gen_end = [[] for _ in range(8)], [[] for _ in range(8)], [[] for _ in range(8)], [[] for _ in range(8)]
for k in np.unique(key8):
    idx = np.where(key8 == k)[0]
    if len(idx) < 4:
        continue
    idx = idx[np.argsort(t[idx, 0])]
    for r, b in enumerate(idx[:8]):
        g = r // 2
        gen_tot[g].append(t[b, 9] - t[b, 0]); gen_pro[g].append(d[b, 0]); gen_start[g].append(t[b, 0] - t0); gen_end[g].append(t[b, 9] - t0)
print("by generation on a CU (two workgroups per generation): workgroup total / prologue / start / end, cycles (mean)")
for g in range(8):
    if gen_tot[g]:
        print(f"  generation {g}: total {np.mean(gen_tot[g]):9.0f}  prologue {np.mean(gen_pro[g]):8.0f}  start {np.mean(gen_start[g]):9.0f}  end {np.mean(gen_end[g]):9.0f}  (n={len(gen_tot[g])})")
span = t[:, 9].max() - t0
print(f"kernel span {span} cycles; a CU's last workgroup ends at {np.mean([max(v) for v in [gen_end[g] for g in range(8) if gen_end[g]]]):.0f} on average")

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
reader = lib.dce_debug_trace_read if os.environ.get('DCE_CONV') == 'direct' else lib.dce_debug_trace_read_wino
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

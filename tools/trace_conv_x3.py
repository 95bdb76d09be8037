#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of conv_x3_kernel (needs build_variant('trace', ['-DDCE_TRACE=1']) and DCE_LIB)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth, _lib
B = 4096
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32_split"
m = contact_cnn(device=0, max_batch=B, precision=PREC); m.load_state_dict(synth.make_state_dict(1)).eval()
x = torch.randn((B, 150, 54), device="cuda")
for _ in range(3): m.predict(x)
torch.cuda.synchronize()
print(m.last_plan())
lib = _lib.load()
buf = np.zeros((B, 16), np.uint64)
assert lib.dce_debug_trace_read_x3(buf.ctypes.data_as(C.c_void_p), B) == 0
t = buf[:, :10].astype(np.int64)
d = np.diff(t, axis=1)
names = ["prologue", "conv1", "store1", "conv2", "store2", "conv3", "store3", "conv4", "feat out"]
print("phase durations (cycles; wave 0 of every workgroup): mean / p10 / p90")
for i, nme in enumerate(names):
    print(f"  {nme:9s} {d[:, i].mean():9.0f} {np.percentile(d[:, i], 10):9.0f} {np.percentile(d[:, i], 90):9.0f}")
tot = t[:, 9] - t[:, 0]
print("block total", tot.mean(), " kernel span", t[:, 9].max() - t[:, 0].min(), " MFMA-phase share", (d[:, [1, 3, 5, 7]].sum(1) / tot).mean())
print("pure MFMA cycles per wave: conv1/2/3 5760 each, conv4 11520 (60 MFMAs x 16 cycles x 6 / 12 K-steps); half of that on two-term operands")

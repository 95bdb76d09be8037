#!/usr/bin/env python3
"""Debug: phase 3 (conv2 X + features Y- + prologue Y) of conv_x3p_kernel, every wave of every workgroup: clock at the K-step boundaries, the tail, the barrier
(build_variant('trace2', ['-DDCE_TRACE=2']), DCE_LIB=deep_contact_estimator_amd/libdce_trace2.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth, _lib
B = 4096
m = contact_cnn(device=0, max_batch=B, precision="bf16_fc"); m.load_state_dict(synth.make_state_dict(1)).eval()
x = torch.randn((B, 150, 54), device="cuda")
for _ in range(3): m.predict(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((256, 8, 16), np.uint64)
assert lib.dce_debug_trace_read_x3p_k(buf.ctypes.data_as(C.c_void_p), 256) == 0
t = buf.astype(np.int64)
t0 = t[:, :, 0].min(axis=1)[:, None, None]          # first wave into conv1
r = t - t0
names = ["enter", "step0", "step1", "step2", "step3", "step4", "step5", "steps done", "-", "behind barrier"]
print("cycles since the workgroup's first wave entered the phase; mean over 256 workgroups, per wave 0..7 (waves w, w+4 share a SIMD)")
for k, nme in enumerate(names):
    print(f"  {nme:18s} " + " ".join(f"{r[:, w, k].mean():7.0f}" for w in range(8)))
d = np.diff(r[:, :, :10], axis=2)
print("durations (mean over workgroups and waves):", " ".join(f"{names[k + 1]}: {d[:, :, k].mean():.0f}" for k in range(9)))

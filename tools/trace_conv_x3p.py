#!/usr/bin/env python3
"""Debug: per-workgroup layer timeline of conv_x3p_kernel -- wave 0's clock at the start of every layer of its last window and in
front of the barrier behind it (needs build_variant('trace', ['-DDCE_TRACE=1']) and DCE_LIB=deep_contact_estimator_amd/libdce_trace.so)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth, _lib
B = int(os.environ.get("TRACE_B", "4096"))
prec = os.environ.get("TRACE_PRECISION", "bf16_fc")
m = contact_cnn(device=0, max_batch=B, precision=prec); m.load_state_dict(synth.make_state_dict(1)).eval()
x = torch.randn((B, 150, 54), device="cuda")
for _ in range(3): m.predict(x)
torch.cuda.synchronize()
print(m.last_plan())
lib = _lib.load()
nb = min(256, B)
buf = np.zeros((nb, 16), np.uint64)
assert lib.dce_debug_trace_read_x3p(buf.ctypes.data_as(C.c_void_p), nb) == 0
t = buf.astype(np.int64)
names = ["conv1", "conv2+pool", "conv3", "conv4+prologue"]
pure = [2880, 2880, 2880, 5760]
print("cycles, mean over workgroups (wave 0, last window): work = layer start -> in front of its barrier; wait = barrier; MFMAs alone")
for l in range(4):
    work = t[:, 2 * l + 1] - t[:, 2 * l]
    wait = (t[:, 2 * l + 2] - t[:, 2 * l + 1]).mean() if l < 3 else float("nan")
    print(f"  {names[l]:15s} work {work.mean():8.0f} (p10 {np.percentile(work, 10):7.0f} p90 {np.percentile(work, 90):7.0f})  wait {wait:7.0f}   2 waves x {pure[l]} = {2 * pure[l]}")
print(f"  one window, conv1 start -> conv4 end: {(t[:, 7] - t[:, 0]).mean():.0f} cycles; MFMAs of one window on a SIMD: 28,800")

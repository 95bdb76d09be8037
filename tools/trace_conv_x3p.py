#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of conv_x3p_kernel -- wave 0's clock at the start of each of the eight phases of its last
window pair and in front of the barrier behind it (needs build_variant('trace', ['-DDCE_TRACE=1']) and
DCE_LIB=deep_contact_estimator_amd/libdce_trace.so)."""
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
nb = min(256, (B + 1) // 2)
buf = np.zeros((nb, 16), np.uint64)
assert lib.dce_debug_trace_read_x3p(buf.ctypes.data_as(C.c_void_p), nb) == 0
t = buf.astype(np.int64)
names = ["1 conv1 X + store3 Y-", "2 conv4 Y- + store1 X", "3 conv2 X + feat Y- + prologue Y", "4 conv1 Y + store2 X", "5 conv3 X + store1 Y",
         "6 conv2 Y + store3 X", "7 conv4 X + store2 Y", "8 conv3 Y + feat X + prologue X+"]
pure = [5760, 11520, 5760, 5760, 5760, 5760, 11520, 5760]
print("cycles, mean over workgroups (wave 0, last window pair): work = phase start -> in front of its barrier; wait = barrier; the SIMD's MFMAs alone")
for l in range(8):
    work = t[:, 2 * l + 1] - t[:, 2 * l]
    wait = (t[:, 2 * l + 2] - t[:, 2 * l + 1]).mean() if l < 7 else float("nan")
    print(f"  {names[l]:34s} work {work.mean():8.0f} (p10 {np.percentile(work, 10):7.0f} p90 {np.percentile(work, 90):7.0f})  wait {wait:7.0f}   {pure[l]}")
print(f"  phases 4..8 + the next 1..3 make a pair; phase 1 start -> phase 8 end of this trace: {(t[:, 15] - t[:, 0]).mean():.0f} cycles; MFMAs of two windows on a SIMD: 57,600")

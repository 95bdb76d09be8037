#!/usr/bin/env python3
"""Debug: per-workgroup phase timeline of conv_x3p_kernel -- for both groups of every workgroup, the start of each phase and the
end of its work (in front of the barrier), last window pair (needs build_variant('trace', ['-DDCE_TRACE=1']) and
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
buf = np.zeros((nb, 64), np.uint64)
assert lib.dce_debug_trace_read_x3p(buf.ctypes.data_as(C.c_void_p), nb) == 0
t = buf.astype(np.int64).reshape(nb, 2, 32)[:, :, :16]
names = ["0 out+in", "1 conv1", "2 store", "3 conv2", "4 store+pool", "5 conv3", "6 store", "7 conv4"]
print("cycles, mean over workgroups (wave 0 of each group, its last window): work = phase start -> in front of the barrier; wait = barrier")
for g in range(2):
    print(f" group {g}")
    for ph in range(8):
        work = t[:, g, 2 * ph + 1] - t[:, g, 2 * ph]
        nxt = t[:, g, 2 * ph + 2] if ph < 7 else None
        wait = (nxt - t[:, g, 2 * ph + 1]).mean() if nxt is not None else float("nan")
        print(f"  {names[ph]:13s} work {work.mean():8.0f} (p10 {np.percentile(work, 10):7.0f} p90 {np.percentile(work, 90):7.0f})   wait {wait:8.0f}")
    print(f"  phases 0..7 of one window: {(t[:, g, 15] - t[:, g, 0]).mean():.0f} cycles to the end of conv4's work")
print("MFMAs of two windows on a SIMD: 57,600 cycles; per wave: conv1/2/3 5,760 each, conv4 11,520")

#!/usr/bin/env python3
"""Debug: phase timeline of the split-bf16 fc.0 GEMM (needs build_variant('x3trace', ['-DX3_TRACE=1']) and DCE_LIB)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth, _lib
m = contact_cnn(device=0, max_batch=4096, precision="fp32_split"); m.load_state_dict(synth.make_state_dict(1))
x = torch.randn((4096, 150, 54), device="cuda")
for _ in range(3): m.predict(x)
torch.cuda.synchronize()
print(m.last_plan())
lib = _lib.load()
buf = np.zeros((8, 32, 4), np.uint64)
assert lib.dce_debug_x3_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)
stride = int(os.environ.get("X3_TRACE_STRIDE", "1"))
print("K-tile start times of wave 0 (ticks, every %d-th K-tile), differences / stride:" % stride)
print((np.diff(t[0, :, 0]) // stride).tolist())
for w in (0, 1, 4, 5):
    tw = t[w, 4:28]
    load = tw[:, 1] - tw[:, 0]; wait1 = tw[:, 2] - tw[:, 1]; math = tw[:, 3] - tw[:, 2]; wait2 = (np.roll(tw[:, 0], -1) - tw[:, 3]) if stride == 1 else np.zeros_like(load)
    print(f"wave {w}: load+waits {load.mean():.0f}  barrier-after-load {wait1.mean():.0f}  math {math.mean():.0f}  barrier-after-math(+issue) {wait2[:-1].mean():.0f}  "
          f"period {np.diff(tw[:, 0]).mean() / stride:.0f}")

#!/usr/bin/env python3
"""Debug: phase timeline of the phased fc.0 GEMM (needs build_variant('phtrace', ['-DPH_TRACE=1']) and DCE_LIB)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth, _lib
# DCE_TRACE_PRECISION=bf16_fc traces the bf16 GEMM; DCE_TUNE=phased_min_tiles1=100000 keeps fc.3 off the phased kernel so that the
# trace left behind is fc.0's
m = contact_cnn(device=0, max_batch=4096, precision=os.environ.get("DCE_TRACE_PRECISION", "fp32")); m.load_state_dict(synth.make_state_dict(1))
x = torch.randn((4096, 150, 54), device="cuda")
for _ in range(3): m.predict(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((8, 64, 4), np.uint64)
assert lib.dce_debug_phase_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)
# NB: the LAST kernel that wrote the trace is fc.3's (128x64 tile, 32 K-tiles); fc.0 overwritten -> report what is there
for w in (0, 1, 4, 5):
    tw = t[w, 4:28]
    load = tw[:, 1] - tw[:, 0]; wait1 = tw[:, 2] - tw[:, 1]; math = tw[:, 3] - tw[:, 2]; wait2 = np.roll(tw[:, 0], -1) - tw[:, 3]
    print(f"wave {w}: load {load.mean():.0f}  barrier-after-load {wait1.mean():.0f}  math {math.mean():.0f}  barrier-after-math {wait2[:-1].mean():.0f}  "
          f"period {np.diff(tw[:, 0]).mean():.0f}")
print("math-end times of waves 0..3 relative to wave 0 (tile 10):", (t[0:4, 10, 3] - t[0, 10, 3]).tolist())
print("math-end times of waves 4..7 relative to wave 4 (tile 10):", (t[4:8, 10, 3] - t[4, 10, 3]).tolist())

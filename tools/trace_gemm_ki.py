#!/usr/bin/env python3
"""Debug: phase timeline of fc_gemm_ki_kernel (needs build_variant('phtrace', ['-DPH_TRACE=1']), DCE_LIB)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth, _lib
os.environ.setdefault("DCE_TUNE", "phased_min_tiles1=100000,gemm_ki=1")       # keeps fc.3 off the phased kernels: the trace left behind is fc.0's
m = contact_cnn(device=0, max_batch=4096, precision="bf16_fc"); m.load_state_dict(synth.make_state_dict(1))
x = torch.randn((4096, 150, 54), device="cuda")
for _ in range(3): m.predict(x)
torch.cuda.synchronize()
print(m.last_plan())
lib = _lib.load()
buf = np.zeros((8, 64, 4), np.uint64)
assert lib.dce_debug_phase_trace_read(buf.ctypes.data_as(C.c_void_p)) == 0
t = buf.astype(np.int64)
for w in (0, 1, 4, 5):
    own = np.arange(8 + (w >> 2), 40, 2)
    tw = t[w, own]
    load = tw[:, 1] - tw[:, 0]; wait1 = tw[:, 2] - tw[:, 1]; math = tw[:, 3] - tw[:, 2]; wait2 = tw[1:, 0] - tw[:-1, 3]
    print(f"wave {w}: load {load.mean():.0f}  barrier-after-load {wait1.mean():.0f}  math {math.mean():.0f}  barrier-after-math {wait2.mean():.0f}  period per own tile (= 2 K-tiles) {np.diff(tw[:, 0]).mean():.0f}")
k = t[0, 62]
print(f"wave 0: kernel start -> loop end {k[1] - k[0]}  exchange {k[2] - k[1]}  bias + stores issued {k[3] - k[2]}  (cycles); first own tile's phase starts {t[0, 0, 0] - k[0]} after kernel start")

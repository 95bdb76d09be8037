#!/usr/bin/env python3
"""Debug: phase timeline of the one-window conv kernel (conv_wino1_kernel; needs the trace variant:
build_variant('trace', ['-DDCE_TRACE=1']), DCE_LIB pointing at it, DCE_TUNE=trace_wino1=1,winoh_max=0,winoq_max=0
so that the one-window kernel runs instead of its half- / quarter-window forms, which carry no marks)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import contact_cnn, synth, _lib
import torch
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
m = contact_cnn(device=0, max_batch=256); m.load_state_dict(synth.make_state_dict(1)).eval()
x = torch.from_numpy(np.random.default_rng(0).standard_normal((B, 150, 54), dtype=np.float32)).cuda()
for _ in range(5): m.predict(x)
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((B, 16), np.uint64)
assert lib.dce_debug_trace_read_wino(buf.ctypes.data_as(C.c_void_p), B) == 0
t = buf[:, :10].astype(np.int64); d = np.diff(t, axis=1)
names = ["prologue", "conv1", "store1", "conv2", "store2+pads", "conv3", "store3", "conv4", "store feat"]
print(f"{B} windows; cycles per phase: mean / min / max   (s_memtime ticks; 100 MHz if the total is ~3.7k, core clock if ~85k)")
for i, nme in enumerate(names):
    print(f"  {nme:12s} {d[:, i].mean():9.0f} {d[:, i].min():9d} {d[:, i].max():9d}")
print("  total       ", (t[:, 9] - t[:, 0]).mean())

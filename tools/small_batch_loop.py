#!/usr/bin/env python3
"""A fixed number of model.predict calls at the reference's batch sizes (1 and 30) and two mid-size batches, for
rocprofv3 --kernel-trace --stats (profiles/*_small_batch_kernel_stats.csv)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
m = contact_cnn(device=0, max_batch=1024, precision=os.environ.get("DCE_LAT_PRECISION", "fp32")); m.load_state_dict(synth.make_state_dict(1))
seq = torch.from_numpy(synth.make_sequence(1024 + 149, 2).astype(np.float32)).cuda()
x = m.zscore_windows(seq)
for B in (int(a) for a in (sys.argv[1:] or ["30"])):
    xb = x[:B].contiguous()
    for _ in range(500):
        m.predict(xb)
    torch.cuda.synchronize()

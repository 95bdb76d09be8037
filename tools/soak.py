#!/usr/bin/env python3
"""Soak (run on the GPU box): 100k+ online pushes checked bit for bit against the sequence path across
resets and buffer compactions, 20,000 bench-sized steps checked for bit stability, 300 context
create/use/destroy cycles checked for device-memory leaks."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from deep_contact_estimator_amd import contact_cnn, synth
free0 = torch.cuda.mem_get_info()[0]
m = contact_cnn(device=0, max_batch=4096); m.load_state_dict(synth.make_state_dict(1, "uniform"))
seq = synth.make_sequence(150 + 9000, 5).astype(np.float32)
ref = m.infer_sequence(seq)
t0 = time.time(); n_ok = 0
for rep in range(12):                       # ~110k pushes, several compactions and resets
    m.online_reset()
    for t in range(len(seq)):
        r = m.online_push(seq[t])
        if r is not None:
            if not np.array_equal(r[0], ref["logits"][t - 149]): raise SystemExit(f"mismatch rep {rep} t {t}")
            n_ok += 1
dt = time.time() - t0
x = m.zscore_windows(torch.from_numpy(seq[:4096 + 149]).cuda())
first = m.predict(x)["logits"].clone()
for i in range(20000):
    out = m.predict(x)
torch.cuda.synchronize()
same = bool(torch.equal(out["logits"], first))
sd2 = synth.make_state_dict(2)
marks = {}
for i in range(300):                        # create/destroy cycles (online buffers, staging and graph-free paths included)
    c = contact_cnn(device=0, max_batch=256); c.load_state_dict(sd2); c.predict(np.zeros((3,150,54),np.float32))
    c.infer_sequence(seq[:400]); [c.online_push(seq[t]) for t in range(152)]; c.close()
    if i in (9, 299): torch.cuda.synchronize(); marks[i] = torch.cuda.mem_get_info()[0]
del x, first, out
m.close(); torch.cuda.empty_cache()
free1 = torch.cuda.mem_get_info()[0]
print(json.dumps({"online_pushes_checked": n_ok, "online_us_per_push": dt / (12 * len(seq)) * 1e6, "predict_20000_steps_bit_stable": same,
                  "ctx_cycles": 300, "leak_MB_between_cycle_10_and_300": (marks[9] - marks[299]) / 1e6,
                  "device_memory_delta_MB_start_to_end": (free0 - free1) / 1e6}))

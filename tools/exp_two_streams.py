#!/usr/bin/env python3
"""Experiment: does running two independent kernel sequences on two streams (so that one's conv
stack overlaps the other's GEMMs on the same CUs) raise total throughput?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth

B = int(os.environ.get("B", 4096))
sd = synth.make_state_dict(1)
ms = [contact_cnn(device=0, max_batch=B) for _ in range(2)]
for m in ms:
    m.load_state_dict(sd)
seq = torch.from_numpy(synth.make_sequence(B + 149, 2).astype(np.float32)).cuda()
x = ms[0].zscore_windows(seq, 0, B)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run(nstreams, steps):
    for _ in range(10):
        for k in range(nstreams):
            with torch.cuda.stream(streams[k]):
                ms[k].predict(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % nstreams
        with torch.cuda.stream(streams[k]):
            ms[k].predict(x)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)

for n in (1, 2, 1, 2):
    print(f"{n} stream(s): {run(n, 200):,.0f} windows/s")

import sys, os
sys.path.insert(0, '/root/repo')
import torch, numpy as np
from deep_contact_estimator_amd import contact_cnn, synth
N = 1_000_000
m = contact_cnn(device=0, max_batch=32768); m.load_state_dict(synth.make_state_dict(1))
g = torch.Generator(device="cuda").manual_seed(3)
seq = torch.randn((N + 149, 54), generator=g, device="cuda")
m.infer_sequence(seq[:40000]); torch.cuda.synchronize()
m.profile(1); m.profile_read(True)
m.infer_sequence(seq); torch.cuda.synchronize()
p = m.profile_read(True)
tot = sum(v["ms"] for v in p.values())
print({k: (round(v["ms"], 2), v["launches"], round(v["ms"] / N * 4096, 4)) for k, v in p.items()}, "total ms", round(tot, 1))
x = m.zscore_windows(seq[:32768 + 149])
m.profile(1); m.profile_read(True)
for _ in range(10): m.predict(x)
torch.cuda.synchronize(); p = m.profile_read(True)
print("non-ZS B=32768 per-4096:", {k: round(v["ms"] / v["launches"] / 8, 4) for k, v in p.items()})

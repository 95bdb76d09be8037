import os, sys
sys.path.insert(0, os.getcwd())
import torch
from deep_contact_estimator_amd import contact_cnn, synth
k32, n = int(sys.argv[1]), int(sys.argv[2])
m = contact_cnn(device=0, max_batch=n, precision="bf16_fc", tune={"bf16_k32": k32}); m.load_state_dict(synth.make_state_dict(1, "uniform")).eval()
x = torch.randn((n, 150, 54), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
for _ in range(12): m.predict(x)
torch.cuda.synchronize()
print(" ".join(m.last_plan()))

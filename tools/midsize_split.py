import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from deep_contact_estimator_amd import contact_cnn, synth
def run(envmin):
    m = contact_cnn(device=0, max_batch=4096, precision="fp32_split", tune=None if envmin is None else {"x3_conv_min": envmin}); m.load_state_dict(synth.make_state_dict(1))
    seq = torch.from_numpy(synth.make_sequence(4096 + 149, 2).astype(np.float32)).cuda()
    x = m.zscore_windows(seq)
    res = {}
    for B in (128, 256, 384, 512, 768, 1024, 1536, 2048, 2560):
        xb = x[:B].contiguous()
        for _ in range(20): m.predict(xb)
        torch.cuda.synchronize()
        n = 200; t0 = time.perf_counter()
        for _ in range(n): m.predict(xb)
        torch.cuda.synchronize()
        res[B] = round((time.perf_counter() - t0) / n * 1e6, 1)
        plan = m.last_plan()[0]
        res[str(B) + "_conv"] = plan
    m.close()
    return res
a = run(None); b = run(1)
for B in (128, 256, 384, 512, 768, 1024, 1536, 2048, 2560):
    print(B, "fp32 conv", a[B], a[str(B) + "_conv"], "| conv_x3_f32", b[B], b[str(B) + "_conv"])

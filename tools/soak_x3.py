#!/usr/bin/env python3
"""Soak of the fp32_split and bf16_fc precisions -- or of the precisions named on the command line (fp32_f16x2) -- (run on the GPU box): 10,000 bench-sized steps each, checked for bit
stability against the first; 1e6-window streaming calls repeated; 200 context create / use / destroy cycles per precision
checked for device-memory leaks (the modes own extra buffers: three-plane features, split weights)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from deep_contact_estimator_amd import contact_cnn, synth
res = {}
sd = synth.make_state_dict(1, "uniform")
seq = torch.from_numpy(synth.make_sequence(4096 + 149, 5).astype(np.float32)).cuda()
for prec in (sys.argv[1:] or ("fp32_split", "bf16_fc")):
    free0 = torch.cuda.mem_get_info()[0]
    m = contact_cnn(device=0, max_batch=4096, precision=prec); m.load_state_dict(sd)
    x = m.zscore_windows(seq)
    first = m.predict(x)["logits"].clone()
    t0 = time.time()
    for i in range(10000):
        out = m.predict(x)
    torch.cuda.synchronize()
    dt = time.time() - t0
    same = bool(torch.equal(out["logits"], first))
    big = torch.randn((200_000 + 149, 54), generator=torch.Generator(device="cuda").manual_seed(1), device="cuda")
    s0 = m.infer_sequence(big)["logits"].clone()
    stable = all(bool(torch.equal(m.infer_sequence(big)["logits"], s0)) for _ in range(5))
    marks = {}
    for i in range(200):
        c = contact_cnn(device=0, max_batch=4096, precision=prec); c.load_state_dict(sd)
        c.predict(x[:3000]); c.predict(x[:200]); c.close()
        if i in (9, 199): torch.cuda.synchronize(); marks[i] = torch.cuda.mem_get_info()[0]
    m.close(); del x, first, out, big, s0; torch.cuda.empty_cache()
    res[prec] = {"predict_10000_steps_bit_stable": same, "windows_per_s_over_the_soak": 4096 * 10000 / dt,
                 "streaming_200k_x5_bit_stable": stable, "ctx_cycles": 200,
                 "leak_MB_between_cycle_10_and_200": (marks[9] - marks[199]) / 1e6,
                 "device_memory_delta_MB_start_to_end": (free0 - torch.cuda.mem_get_info()[0]) / 1e6}
print(json.dumps(res))

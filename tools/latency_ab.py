import json, os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch, ctypes as C
from deep_contact_estimator_amd import contact_cnn, synth
sd = synth.make_state_dict(1, "uniform")
seq = synth.make_sequence(150 + 2400, seed=2).astype(np.float32)
def trace(m):
    t = (C.c_uint64 * 16)()
    if m._lib.dce_debug_latency_trace(m._ctx, t) != 0: return None
    t = list(t); base = min(v for v in t if v)
    return [round((v - base) / 100.0, 2) if v else None for v in t]
for tr in (0, 1):
    if tr: os.environ["DCE_LAT_TRACE"] = "1"
    for delay in (0, 100, 300, 600):
        m = contact_cnn(device=0, max_batch=64, tune={"latency": 1, "latency_fc_delay": delay}); m.load_state_dict(sd).eval()
        x = m.zscore_windows(torch.from_numpy(seq[:150 + 63]).cuda())
        xb = x[:1].contiguous()
        for _ in range(50): m.predict(xb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(1000): m.predict(xb)
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / 1000 * 1e6
        m.online_reset()
        for t in range(350): m.online_push(seq[t])
        t0 = time.perf_counter()
        for t in range(350, 2350): m.online_push(seq[t])
        push = (time.perf_counter() - t0) / 2000 * 1e6
        print(json.dumps({"trace": tr, "fc_delay_ticks": delay, "predict_1_us": round(us, 2), "push_us": round(push, 2), "stamps": trace(m) if tr else None}), flush=True)
        m.close()

"""Development check of the micro-batch latency kernel (latency_mb.hip): parity against the oracle and time per call."""
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
from oracle import oracle as orc

sd = synth.make_state_dict(1, "uniform")
m = contact_cnn(device=0, max_batch=64, tune={"latency": 1}); m.load_state_dict(sd).eval()
b = contact_cnn(device=0, max_batch=64); b.load_state_dict(sd).eval()
o = orc.Oracle(sd)
rng = np.random.default_rng(3)
res = {}
for n in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,3,8,15,16,17,30,31,32,33").split(",")]:
    worst, flips = 0.0, 0
    for rep in range(6):
        x = rng.standard_normal((n, 150, 54), dtype=np.float32) * (1.0 + rep)
        out = m.predict(x)
        plan = m.last_plan()
        ref = o.forward_windows(x)
        err = np.abs(out["logits"] - ref["logits"])
        bound = 1e-5 * np.abs(ref["logits"]).max() + 1e-4 * np.abs(ref["logits"])
        worst = max(worst, float((err / bound).max()))
        flips += int((out["pred"] != ref["pred"]).sum())
        assert np.array_equal(out["contacts"], ((out["pred"][:, None] & np.array([8, 4, 2, 1])) != 0).astype(np.uint8))
    seq = synth.make_sequence(150 + n - 1, seed=n).astype(np.float32)
    so, sr = m.infer_sequence(seq), o.infer_sequence(seq)
    plan_s = m.last_plan()
    es = float((np.abs(so["logits"] - sr["logits"]) / (1e-5 * np.abs(sr["logits"]).max() + 1e-4 * np.abs(sr["logits"]))).max())
    xd = torch.from_numpy(x).cuda()
    def timeit(mod, k=300):
        for _ in range(30): mod.predict(xd)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): mod.predict(xd)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
    tr = None
    if os.environ.get("DCE_LAT_TRACE"):
        import ctypes as C
        m.predict(xd); torch.cuda.synchronize()
        st = (C.c_ulonglong * 16)()
        m._lib.dce_debug_latency_trace(m._ctx, st)
        t0 = st[1]
        tr = {k: round((st[i] - t0) / 100.0, 2) for i, k in ((2, "conv_done"), (3, "conv_posted"), (4, "fc0_w_requested"), (5, "fc0_w_landed"), (6, "feat_seen"), (7, "h1_posted"),
                                                            (8, "h1_seen"), (9, "fc3_posted"), (10, "partials_seen"), (11, "done"), (14, "conv_w0_stores_acked"), (15, "conv_all_waves"))}
    def dist(mod, k=200):
        ts = []
        for _ in range(k):
            torch.cuda.synchronize(); t0 = time.perf_counter(); mod.predict(xd); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e6)
        ts.sort(); return [round(ts[int(q * (k - 1))], 1) for q in (0.0, 0.5, 0.9, 0.99, 1.0)]
    res[n] = {"trace_us": tr, "synced_call_us_min_p50_p90_p99_max": dist(m), "plan": plan, "plan_seq": plan_s, "err_over_bound": round(worst, 4), "seq_err_over_bound": round(es, 4), "argmax_diff": flips,
              "us_latency_mode": round(timeit(m), 2), "us_batch_path": round(timeit(b), 2)}
    print(n, json.dumps(res[n]), flush=True)

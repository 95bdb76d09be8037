#!/usr/bin/env python3
"""Randomised differential run of the product path against the ORACLE: random call sizes (on both sides of every plan threshold), precisions, entry points
(pre-normalised windows | raw rows with the fused z-score), host | device pointers, latency option, checkpoints (He-uniform seeds, scaled layers) and input
scales; one context per case, several calls per context (sizes interleaved, so that stale buffers / flags of a larger call meet a smaller one).  Contract per
case: fp32 / fp32_f16x2: logits within 1e-5 max|ref| + 1e-4 |ref|, argmax equal where the reference's top-2 margin exceeds 1e-3 max|logit|, contacts = bits(pred);
bf16_fc: within 6e-3 of the largest logit of the MODE'S OWN restatement (Oracle(bf16_fc=True): features and h1 rounded to bf16, bf16 weights, fp32 sums -- the
band covers rounding-boundary flips of single bf16 entries), argmax equal above a 1e-2 margin; its distance to the fp32 oracle is reported, not judged.  Prints one JSON summary; exit code 1 on a violation.

    python tools/fuzz_parity.py [seconds, default 300] [seed]        (profiles/r6p_fuzz_parity.json)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deep_contact_estimator_amd import contact_cnn, synth
from oracle import oracle as orc

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
EDGES = [1, 2, 3, 8, 9, 12, 13, 16, 17, 30, 31, 32, 33, 64, 65, 128, 129, 256, 257, 640, 641, 1024, 1280, 1281, 2048, 2049, 2816, 2817, 3000, 4096, 4100, 5000]


def pick_size():
    r = rng.random()
    if r < 0.6:
        return int(rng.choice(EDGES))
    if r < 0.85:
        return int(rng.integers(1, 80))
    return int(rng.integers(80, 5200))


cases, worst, viol, plans, info = 0, {}, [], {}, {}
t_end = time.time() + budget
while time.time() < t_end:
    precision = str(rng.choice(["fp32", "fp32", "fp32_f16x2", "bf16_fc"]))
    latency = precision == "fp32" and rng.random() < 0.4
    ck_seed = int(rng.integers(1, 50))
    sd = synth.make_state_dict(ck_seed, "uniform")
    if rng.random() < 0.3:                                    # a layer scaled by a power of two and its successor's inverse: other activation ranges, same function up to rounding
        k = str(rng.choice(["block1.0", "block1.2", "block2.0", "block2.2", "fc.0"]))
        f = float(2.0 ** int(rng.integers(-6, 7)))
        sd = {n: (v * f if n.startswith(k + ".") else v) for n, v in sd.items()}
    sizes = [pick_size() for _ in range(int(rng.integers(2, 5)))]
    in_scale = float(10.0 ** rng.uniform(-2, 2)) if rng.random() < 0.4 else 1.0
    kind = str(rng.choice(["normal", "ar1"]))
    tune = {"latency": 1} if latency else None
    m = contact_cnn(device=0, max_batch=int(rng.choice([64, 1024, 4096, 8192])), precision=precision, tune=tune)
    m.load_state_dict(sd).eval()
    o = orc.Oracle(sd, bf16_fc=True) if precision == "bf16_fc" else orc.Oracle(sd)
    o32 = orc.Oracle(sd) if precision == "bf16_fc" else None
    for n in sizes:
        seq = (synth.make_sequence(149 + n, int(rng.integers(1, 10 ** 6)), kind) * in_scale).astype(np.float32)
        raw = rng.random() < 0.5
        dev = rng.random() < 0.5
        zw = orc.zscore_windows(seq)
        if not raw and rng.random() < 0.3:
            zw = (zw * np.float32(10.0 ** rng.uniform(-1.5, 1.5))).astype(np.float32)      # pre-normalised windows need not be z-scores
        ref = o.forward_windows(zw)
        if raw:
            x = torch.from_numpy(seq).cuda() if dev else seq
            got = m.infer_sequence(x)
        else:
            x = torch.from_numpy(zw).cuda() if dev else zw
            got = m.predict(x)
        plan = " ".join(m.last_plan())
        lg, pr, ct = [np.asarray(got[k].cpu() if dev else got[k]) for k in ("logits", "pred", "contacts")]
        big = float(np.abs(ref["logits"]).max())
        err = np.abs(lg.astype(np.float64) - ref["logits"])
        srt = np.sort(ref["logits"], axis=1); margin = srt[:, -1] - srt[:, -2]
        if precision == "bf16_fc":
            score = float(err.max() / big) / 6e-3
            r32 = o32.forward_windows(zw)["logits"]
            info["bf16_fc_vs_fp32_oracle_max_err_over_largest_logit"] = max(info.get("bf16_fc_vs_fp32_oracle_max_err_over_largest_logit", 0.0), float(np.abs(lg - r32).max() / np.abs(r32).max()))
            safe = margin > 1e-2 * big
        else:
            score = float((err / (1e-5 * big + 1e-4 * np.abs(ref["logits"]))).max())
            safe = margin > 1e-3 * big
        ok = score <= 1.0 and np.array_equal(pr[safe], ref["pred"][safe]) and np.array_equal(ct, orc.decimal2binary(pr)) and np.isfinite(lg).all()
        key = precision + ("+latency" if latency else "")
        worst[key] = max(worst.get(key, 0.0), score)
        plans.setdefault(key, {}).setdefault(plan, 0)
        plans[key][plan] += 1
        cases += 1
        if not ok:
            viol.append({"precision": key, "n": n, "raw": raw, "device": dev, "plan": plan, "score": score, "ck_seed": ck_seed, "in_scale": in_scale, "kind": kind,
                         "argmax_diff_safe": int((pr[safe] != ref["pred"][safe]).sum())})
            if len(viol) >= 10:
                break
    m.close()
    if len(viol) >= 10:
        break
print(json.dumps({"seconds": budget, "seed": seed, "calls": cases, "violations": viol, "worst_err_over_bound": {k: round(v, 4) for k, v in worst.items()},
                  "distinct_plans": {k: len(v) for k, v in plans.items()}, "info": info, "plans": plans}))
sys.exit(1 if viol else 0)

#!/usr/bin/env python3
"""Randomised parity sweep of the fp32_split precision -- or, with PRECISION=fp32_f16x2, of the two-term fp16 precision, whose trials also scale the
windows and single layers by powers of two far outside fp16's range -- (run on the GPU box): random batch sizes around the mode's thresholds
(127 / 128: the three-term conv stack; 2816 / 2817: the split-bf16 fc.0 GEMM) and inside its ranges, random checkpoints (random gain,
non-zero biases), random input statistics (i.i.d. / AR(1), wild per-channel scales and offsets), streaming (fused z-score) and
materialised-window entry, chunking by max_batch -- every row against the CPU oracle at the fp32 contract."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import contact_cnn, synth
from oracle import oracle as orc

TRIALS = int(os.environ.get("TRIALS", 60))
PRECISION = os.environ.get("PRECISION", "fp32_split")
rng = np.random.default_rng(int(os.environ.get("SEED", 77)))
edges = [100, 127, 128, 129, 255, 256, 700, 1023, 1025, 1280, 1281, 1290, 2048, 2815, 2816, 2817, 2900, 3072, 3333, 4096, 4099, 5003]
worst, flips, total, plans = 0.0, 0, 0, {}
t0 = time.time()
for trial in range(TRIALS):
    seed = int(rng.integers(1, 1 << 30))
    sd = synth.make_state_dict(seed, "uniform")
    gain = float(rng.uniform(0.6, 1.6))
    sd = {k: (v * gain if k.endswith("weight") else v) for k, v in sd.items()}
    n = int(rng.choice(edges)) if rng.random() < 0.7 else int(rng.integers(128, 6000))
    kind = "ar1" if rng.random() < 0.5 else "normal"
    seq = synth.make_sequence(n + 149, int(rng.integers(0, 1 << 30)), kind).astype(np.float32)
    if rng.random() < 0.3:
        seq = seq * (10.0 ** rng.uniform(-3, 3, 54)).astype(np.float32) + rng.uniform(-100, 100, 54).astype(np.float32)
    if PRECISION == "fp32_f16x2" and rng.random() < 0.5:
        # a layer's weights times 2^e, the next layer's times 2^-e (biases with the activations they are added to): the activations between
        # them swing over up to 80 binades; the z-score entry keeps the input itself at O(1)
        l = int(rng.integers(0, 5)); e = int(rng.integers(-40, 41))
        names = ["block1.0", "block1.2", "block2.0", "block2.2", "fc.0", "fc.3"]
        sd = dict(sd)
        sd[names[l] + ".weight"] = (sd[names[l] + ".weight"] * np.float32(2.0 ** e)).astype(np.float32)
        sd[names[l] + ".bias"] = (sd[names[l] + ".bias"] * np.float32(2.0 ** e)).astype(np.float32)
        sd[names[l + 1] + ".weight"] = (sd[names[l + 1] + ".weight"] * np.float32(2.0 ** -e)).astype(np.float32)
    mb = int(rng.choice([3000, 4096, 8192]))
    m = contact_cnn(device=0, max_batch=mb, precision=PRECISION)
    m.load_state_dict(sd).eval()
    ref = orc.Oracle(sd).infer_sequence(seq)
    a = m.infer_sequence(seq)
    pa = tuple(m.last_plan())
    b = m.predict(m.zscore_windows(seq))
    plans[pa[0] + "+" + pa[1]] = plans.get(pa[0] + "+" + pa[1], 0) + 1
    scale = np.abs(ref["logits"]).max()
    for tag, o in (("stream", a), ("windows", b)):
        bound = 1e-5 * scale + 1e-4 * np.abs(ref["logits"])
        r = float((np.abs(o["logits"] - ref["logits"]) / bound).max())
        worst = max(worst, r)
        assert r <= 1.0, (trial, tag, n, seed, r)
        srt = np.sort(ref["logits"], axis=1)
        clear = (srt[:, -1] - srt[:, -2]) > 1e-3 * scale
        assert np.array_equal(o["pred"][clear], ref["pred"][clear]), (trial, tag, "argmax")
        flips += int((o["pred"] != ref["pred"]).sum()); total += n
        bits = ((o["pred"][:, None] >> np.array([3, 2, 1, 0])) & 1).astype(np.uint8)
        assert np.array_equal(o["contacts"], bits), (trial, tag, "bits")
    m.close()
print(json.dumps({"precision": PRECISION, "trials": TRIALS, "rows_checked": total, "max_err_over_bound": worst,
                  "sub_margin_argmax_differences": flips, "first_two_kernels_of_the_last_chunk": plans, "seconds": round(time.time() - t0, 1)}))

#!/usr/bin/env python3
"""DCE_BF16_FC with its conv stack on THREE-term operands (six MFMAs per product: fp32-grade features, then rounded to bf16) against
the same stack on TWO-term operands (three MFMAs per product, ~17 significant bits, then rounded to bf16): logits of both against
the fp64-accumulating oracle on the same windows, as |difference| / max|reference logit|, plus argmax differences and how many of
those sit above the noise margin (1e-3 of the largest logit) -- and the two builds against each other.
    python tools/bf16_terms_audit.py [--quick] > profiles/<round>_bf16_terms_audit.json     (GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from deep_contact_estimator_amd import contact_cnn, synth
from oracle import oracle as orc

QUICK = "--quick" in sys.argv
N = 8192 if QUICK else 65536


def model(terms):
    return contact_cnn(device=0, max_batch=32768, precision="bf16_fc", tune={"x3_bf16_terms": terms})


def report(got, ref):
    scale = float(np.abs(ref["logits"]).max())
    d = np.abs(got["logits"].astype(np.float64) - ref["logits"]) / scale
    srt = np.sort(ref["logits"], axis=1)
    margin = srt[:, -1] - srt[:, -2]
    diff = got["pred"] != ref["pred"]
    return {"max_rel_to_largest_logit": float(d.max()), "p999": float(np.percentile(d, 99.9)), "rms": float(np.sqrt((d ** 2).mean())),
            "argmax_differences": int(diff.sum()), "above_noise_margin": int((diff & (margin > 1e-3 * scale)).sum())}


def main():
    sd = synth.make_state_dict(1, "uniform")
    o = orc.Oracle(sd)
    rep = {"protocol": __doc__.split("\n\n")[0], "quick": QUICK, "windows_per_set": N, "sets": {}}
    ms = {}
    for t in (3, 2):
        ms[t] = model(t); ms[t].load_state_dict(sd).eval()
    rng = np.random.default_rng(404)
    sets = {"normal_sequence": rng.standard_normal((N + 149, 54)).astype(np.float32),
            "ar1_sequence": synth.make_sequence(N + 149, 5, "ar1").astype(np.float32),
            "channels_scaled_1e-6_to_1e6": (rng.standard_normal((N + 149, 54)) * 10.0 ** rng.uniform(-6, 6, 54)).astype(np.float32)}
    for name, seq in sets.items():
        ref = o.infer_sequence(seq)
        r = {}
        out = {}
        for t in (3, 2):
            out[t] = ms[t].infer_sequence(seq)
            r[f"terms_{t}"] = dict(report(out[t], ref), plan=ms[t].last_plan()[0])
        scale = float(np.abs(ref["logits"]).max())
        r["terms_2_vs_terms_3"] = {"max_rel_to_largest_logit": float(np.abs(out[2]["logits"].astype(np.float64) - out[3]["logits"]).max() / scale),
                                   "argmax_differences": int((out[2]["pred"] != out[3]["pred"]).sum())}
        rep["sets"][name] = r
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()

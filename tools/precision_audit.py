#!/usr/bin/env python3
"""Precision audit (run on the GPU box): three fp32 evaluations of the path -- the reference's own arithmetic (PyTorch-CPU fp32,
oracle/torch_ref.py), DCE_FP32 (exact fp32 MFMA, one fmaf chain per output) and DCE_FP32_SPLIT (fp32 operands as three bf16 terms on
the bf16 matrix pipe) -- and a fourth that claims the fp32 TOLERANCE, not fp32 operands: DCE_FP32_F16X2 (two fp16 terms of the operand times
a power of two chosen per window and layer, three MFMAs per product) -- each against the fp64-accumulating oracle, as err / bound with bound = 1e-5 max|ref| + 1e-4 |ref| (the
contract of BASELINE.json's "logits within a stated fp32 tolerance"; reference src/contact_cnn.py:60-66, utils/data_handler.py:55-56).

Sets: 1e6 N(0,1) windows and 200k AR(1) windows through the z-score entry (logits of every window); per-layer taps (features,
fc.0, fc.3, logits) on 4096 windows of each; and adversarial sets -- channels scaled 1e-6 .. 1e6, spread / offset = 1e-6,
per-layer weight scales 1e-3 .. 1e3, activations pushed to the fp32 / bf16 subnormal boundary (does the matrix pipe flush the
third term?), inputs in the top binade (bf16's largest finite number is below fp32's: the first term of a split can round to Inf).

    python tools/precision_audit.py [--quick] > profiles/r5_precision_audit.json      (round 5: DCE_FP32_SPLIT with its range guard, and without)
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deep_contact_estimator_amd import contact_cnn, synth
from oracle import oracle as orc
from oracle import torch_ref

QUICK = "--quick" in sys.argv
N_NORMAL, N_AR1, N_ADV = (20_000, 10_000, 4096) if QUICK else (1_000_000, 200_000, 8192)
torch.set_num_threads(os.cpu_count() or 1)


def stats(got, ref):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    bound = 1e-5 * np.abs(ref).max() + 1e-4 * np.abs(ref)
    r = np.abs(got - ref) / bound
    nan_mismatch = int((np.isnan(got) != np.isnan(ref)).sum())
    r = r[np.isfinite(r)]
    return {"max": float(r.max()) if r.size else None, "p999": float(np.percentile(r, 99.9)) if r.size else None,
            "nonfinite_mismatches": nan_mismatch}


def argmax_report(pred, ref_logits, ref_pred):
    srt = np.sort(ref_logits, axis=1)
    margin = srt[:, -1] - srt[:, -2]
    floor = 1e-3 * np.nanmax(np.abs(ref_logits))
    diff = pred != ref_pred
    return {"differences": int(diff.sum()), "above_noise_margin": int((diff & (margin > floor)).sum()),
            "largest_margin_of_a_difference": float(margin[diff].max()) if diff.any() else 0.0, "noise_margin": float(floor)}


def torch_cpu_sequence(sd_t, seq, chunk=8192):
    """The reference's arithmetic: fp32 z-score per window (torch.mean / torch.std, unbiased) + the fp32 model, batched."""
    s = torch.from_numpy(seq)
    n = s.shape[0] - 149
    out = np.empty((n, 16), np.float32)
    for b0 in range(0, n, chunk):
        b1 = min(b0 + chunk, n)
        w = s[b0:b1 + 149].unfold(0, 150, 1).permute(0, 2, 1)              # (b, 150, 54) views
        w = (w - w.mean(dim=1, keepdim=True)) / w.std(dim=1, keepdim=True)
        out[b0:b1] = torch_ref.forward(sd_t, w.contiguous()).numpy()
    return out


def models(sd, max_batch=32768, unguarded=False):
    """fp32, fp32_split (with its range guard: the default) and -- for the adversarial window sets -- fp32_split with the guard switched
    off (option split_guard=0: round 4's behaviour, the rows that show what the guard is for)."""
    ms = {}
    for name, p, tune in (("fp32", "fp32", None), ("fp32_split", "fp32_split", None), ("fp32_f16x2", "fp32_f16x2", None)) + ((("fp32_split_unguarded", "fp32_split", {"split_guard": 0}),) if unguarded else ()):
        ms[name] = contact_cnn(device=0, max_batch=max_batch, precision=p, tune=tune)
        ms[name].load_state_dict(sd).eval()
    return ms


def audit_sequence(name, sd, seq, taps_n=4096):
    t0 = time.time()
    o = orc.Oracle(sd)
    ref = o.infer_sequence(seq)
    sd_t = torch_ref.to_torch(sd)
    res = {"windows": int(seq.shape[0] - 149), "logit_scale": float(np.nanmax(np.abs(ref["logits"]))), "evaluations": {}}
    tl = torch_cpu_sequence(sd_t, seq)
    res["evaluations"]["pytorch_cpu_fp32"] = {"logits": stats(tl, ref["logits"]), "argmax": argmax_report(tl.argmax(1), ref["logits"], ref["pred"])}
    ms = models(sd)
    zw = orc.zscore_windows(seq[:taps_n + 149])
    ref_t = o.forward_windows(zw, taps=True)
    with torch.no_grad():
        tt = {}
        x = torch.from_numpy(zw).permute(0, 2, 1)
        F = torch.nn.functional
        x = F.relu(F.conv1d(x, sd_t["block1.0.weight"], sd_t["block1.0.bias"], padding=1)); x = F.relu(F.conv1d(x, sd_t["block1.2.weight"], sd_t["block1.2.bias"], padding=1))
        x = F.max_pool1d(x, 2, 2)
        x = F.relu(F.conv1d(x, sd_t["block2.0.weight"], sd_t["block2.0.bias"], padding=1)); x = F.relu(F.conv1d(x, sd_t["block2.2.weight"], sd_t["block2.2.bias"], padding=1))
        tt["feat"] = F.max_pool1d(x, 2, 2).reshape(x.shape[0], -1)
        tt["h1"] = F.relu(F.linear(tt["feat"], sd_t["fc.0.weight"], sd_t["fc.0.bias"]))
        tt["h2"] = F.relu(F.linear(tt["h1"], sd_t["fc.3.weight"], sd_t["fc.3.bias"]))
        tt["logits"] = F.linear(tt["h2"], sd_t["fc.6.weight"], sd_t["fc.6.bias"])
    res["evaluations"]["pytorch_cpu_fp32"]["layers"] = {k: stats(tt[k].numpy(), ref_t[k]) for k in ("feat", "h1", "h2", "logits")}
    for p, m in ms.items():
        out = m.infer_sequence(seq)
        e = {"logits": stats(out["logits"], ref["logits"]), "argmax": argmax_report(out["pred"], ref["logits"], ref["pred"]), "plan_of_last_launch": m.last_plan()}
        # per layer: features ARE the conv stack's output; the split mode's chip-filling launch keeps them as three bf16 planes, so
        # its feature tap comes from the 4096-window launch with the tap switched on (fp32 features, split by a kernel of its own)
        t = m.forward_taps(zw)
        e["layers"] = {k: stats(t[k], ref_t[k]) for k in ("feat", "h1", "h2", "logits")}
        e["layers_plan"] = m.last_plan()
        if p == "fp32_f16x2":                                  # the two-term fp16 conv stack itself, layer by layer
            ct = m.conv_layer_taps(zw[:64], "h2")
            lt = [o.layer_taps(w) for w in zw[:8]]
            e["conv_h2_layers_8_windows"] = {k: stats(ct[k][:8], np.stack([t[k] for t in lt])) for k in ("conv1", "conv2", "pool1", "conv3", "conv4")}
            e["conv_h2_layers_8_windows"]["feat_64_windows"] = stats(ct["feat"], ref_t["feat"][:64])
        if p == "fp32_split":                                  # the three-term conv stack itself, layer by layer (the tap takes <= 64 windows)
            ct = m.conv_layer_taps(zw[:64], "x3")
            lt = [o.layer_taps(w) for w in zw[:8]]
            e["conv_x3_layers_8_windows"] = {k: stats(ct[k][:8], np.stack([t[k] for t in lt])) for k in ("conv1", "conv2", "pool1", "conv3", "conv4")}
            e["conv_x3_layers_8_windows"]["feat_64_windows"] = stats(ct["feat"], ref_t["feat"][:64])
        res["evaluations"]["dce_" + p] = e
        m.close()
    res["seconds"] = round(time.time() - t0, 1)
    print(f"[audit] {name}: " + "  ".join(f"{k} {v['logits']['max']:.3f}" for k, v in res["evaluations"].items()), file=sys.stderr, flush=True)
    return res


def audit_windows(name, sd, win, note):
    """Pre-normalised windows straight into the model (no z-score): what the adversarial sets need."""
    o = orc.Oracle(sd)
    ref = o.forward_windows(win)
    sd_t = torch_ref.to_torch(sd)
    tl = torch_ref.forward(sd_t, torch.from_numpy(win)).numpy()
    res = {"windows": int(win.shape[0]), "note": note, "logit_scale": float(np.nanmax(np.abs(ref["logits"]))) if np.isfinite(ref["logits"]).any() else None,
           "oracle_nonfinite_rows": int((~np.isfinite(ref["logits"])).any(1).sum()), "evaluations": {}}
    res["evaluations"]["pytorch_cpu_fp32"] = {"logits": stats(tl, ref["logits"]), "argmax": argmax_report(np.nan_to_num(tl, nan=-np.inf).argmax(1), ref["logits"], ref["pred"])}
    for p, m in models(sd, max_batch=win.shape[0], unguarded=True).items():
        out = m.predict(win)
        res["evaluations"]["dce_" + p] = {"logits": stats(out["logits"], ref["logits"]), "argmax": argmax_report(out["pred"], ref["logits"], ref["pred"]),
                                           "plan": m.last_plan()}
        if p == "fp32_split":
            g = m.split_guard()
            res["evaluations"]["dce_" + p]["range_guard"] = {k: g[k] for k in ("refused", "x_hi", "x_lo", "windows_out_of_range", "fallbacks_run", "reason")}
        m.close()
    print(f"[audit] {name}: " + "  ".join(f"{k} {v['logits']['max']}" for k, v in res["evaluations"].items()), file=sys.stderr, flush=True)
    return res


def main():
    sd = synth.make_state_dict(1, "uniform")
    rng = np.random.default_rng(2026)
    rep = {"protocol": __doc__.split("\n\n")[0], "quick": QUICK, "host_threads": os.cpu_count(), "sets": {}}
    rep["sets"]["normal"] = audit_sequence("normal", sd, rng.standard_normal((N_NORMAL + 149, 54)).astype(np.float32))
    rep["sets"]["ar1"] = audit_sequence("ar1", sd, synth.make_sequence(N_AR1 + 149, 5, "ar1").astype(np.float32))
    # ---- adversarial, through the z-score entry
    base = rng.standard_normal((N_ADV + 149, 54))
    rep["sets"]["channels_scaled_1e-6_to_1e6"] = audit_sequence("channel scales", sd, (base * 10.0 ** rng.uniform(-6, 6, 54)).astype(np.float32), taps_n=min(4096, N_ADV))
    rep["sets"]["spread_over_offset_1e-6"] = audit_sequence("sigma/mu 1e-6", sd, (1.0 + 1e-6 * base).astype(np.float32), taps_n=min(4096, N_ADV))
    # ---- adversarial, pre-normalised windows
    win = rng.standard_normal((N_ADV, 150, 54)).astype(np.float32)
    sd_w = {k: v.copy() for k, v in sd.items()}
    for (wk, bk), f in zip((("block1.0.weight", "block1.0.bias"), ("block1.2.weight", "block1.2.bias"), ("block2.0.weight", "block2.0.bias"),
                            ("block2.2.weight", "block2.2.bias"), ("fc.0.weight", "fc.0.bias"), ("fc.3.weight", "fc.3.bias")), (1e3, 1e-3, 1e3, 1e-3, 1e3, 1e-3)):
        sd_w[wk] = (sd_w[wk] * f).astype(np.float32)
    # biases scaled with the activations they are added to (cumulative scale of the layers in front)
    cum = 1.0
    for (wk, bk), f in zip((("block1.0.weight", "block1.0.bias"), ("block1.2.weight", "block1.2.bias"), ("block2.0.weight", "block2.0.bias"),
                            ("block2.2.weight", "block2.2.bias"), ("fc.0.weight", "fc.0.bias"), ("fc.3.weight", "fc.3.bias")), (1e3, 1e-3, 1e3, 1e-3, 1e3, 1e-3)):
        cum *= f
        sd_w[bk] = (sd_w[bk] * cum).astype(np.float32)
    rep["sets"]["weights_scaled_1e3_1e-3_alternating"] = audit_windows("weight scales", sd_w, win, "layer l's weights x 1e3 / 1e-3 alternating: activations swing over six decades")
    for e in (-100, -110, -118):
        sd_s = {k: v.copy() for k, v in sd.items()}
        sd_s["block1.0.bias"] = (sd_s["block1.0.bias"] * 2.0 ** e).astype(np.float32)
        # the input scaled into the subnormal neighbourhood; conv1's bias with it; conv2's weights scale back up so that the rest of
        # the net sees ordinary numbers: conv1's products and the third terms of its operands (2^-16 of the value) go subnormal
        sd_s["block1.2.weight"] = (sd_s["block1.2.weight"] * 2.0 ** (-e)).astype(np.float32)
        rep["sets"][f"inputs_x_2^{e}"] = audit_windows(f"inputs 2^{e}", sd_s, (win * np.float32(2.0 ** e)).astype(np.float32),
                                                       f"windows x 2^{e} (third terms of the split below 2^{e - 16}: subnormal in bf16 and fp32), conv2 weights x 2^{-e}")
    top = win.copy()
    top *= np.float32(3.0e37)                                  # |x| up to ~1.5e38; a sprinkle of values in the last binade, above bf16's largest finite number
    idx = rng.integers(0, top.size, 2000)
    top.reshape(-1)[idx] = np.float32(3.395e38) * np.sign(top.reshape(-1)[idx])
    sd_t = {k: v.copy() for k, v in sd.items()}
    sd_t["block1.0.weight"] = (sd_t["block1.0.weight"] * 1e-38 / 3.0).astype(np.float32)
    rep["sets"]["inputs_in_the_top_binade"] = audit_windows("top binade", sd_t, top, "windows x 3e37 with 2000 samples at +-3.395e38 (> bf16 max 3.3895e38: the split's first term rounds to Inf), conv1 weights x 3.3e-39 (subnormal-free: 1e-38 / 3)")
    # a set that passes the STATIC guard (ordinary-size weights) and trips the per-window check: inputs x 3e37 with samples at 1.6e38, conv1 x 1e-8
    top2 = (win * np.float32(3.0e37)).astype(np.float32)
    idx = rng.integers(0, top2.size, 2000)
    top2.reshape(-1)[idx] = np.float32(1.6e38) * np.sign(top2.reshape(-1)[idx])
    sd_t2 = {k: v.copy() for k, v in sd.items()}
    sd_t2["block1.0.weight"] = (sd_t2["block1.0.weight"] * np.float32(1e-8)).astype(np.float32)
    rep["sets"]["inputs_x_3e37_conv1_x_1e-8"] = audit_windows("inputs 3e37", sd_t2, top2, "windows x 3e37 with 2000 samples at +-1.6e38 (above the guard's 2^126, below the fp32 Winograd transform's own limit 1.7e38), conv1 weights x 1e-8: passes the static guard, every window fails the per-window check")
    # ---- the rule of VERDICT r3 item 4
    verdict = {}
    for name, s in rep["sets"].items():
        ev = s["evaluations"]
        t, sp = ev["pytorch_cpu_fp32"]["logits"]["max"], ev["dce_fp32_split"]["logits"]["max"]
        ung = ev.get("dce_fp32_split_unguarded")
        h2 = ev["dce_fp32_f16x2"]
        verdict[name] = {"pytorch_cpu": t, "dce_fp32": ev["dce_fp32"]["logits"]["max"], "dce_fp32_split": sp,
                         "dce_fp32_f16x2": h2["logits"]["max"], "f16x2_plan": h2.get("plan") or h2.get("plan_of_last_launch"),
                         "f16x2_nonfinite_mismatches": h2["logits"]["nonfinite_mismatches"],
                         "f16x2_within_the_contract": h2["logits"]["max"] is not None and h2["logits"]["max"] <= 1.0 and h2["logits"]["nonfinite_mismatches"] == 0,
                         "f16x2_above_margin_argmax_differences": h2["argmax"]["above_noise_margin"],
                         "dce_fp32_split_unguarded": ung["logits"]["max"] if ung else None,
                         "split_plan": ev["dce_fp32_split"].get("plan") or ev["dce_fp32_split"].get("plan_of_last_launch"),
                         "split_nonfinite_mismatches": ev["dce_fp32_split"]["logits"]["nonfinite_mismatches"], "fp32_nonfinite_mismatches": ev["dce_fp32"]["logits"]["nonfinite_mismatches"],
                         "split_within_2x_of_pytorch_cpu": (sp is not None and t is not None and sp <= 2.0 * max(t, 1e-9)),
                         "split_within_the_contract": sp is not None and sp <= 1.0 and ev["dce_fp32_split"]["logits"]["nonfinite_mismatches"] == 0,
                         "split_above_margin_argmax_differences": ev["dce_fp32_split"]["argmax"]["above_noise_margin"]}
    rep["summary"] = verdict
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()

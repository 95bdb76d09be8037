#!/usr/bin/env python3
"""Round 6, the last algorithmic lever of the exact-fp32 conv stack PRICED before anything is built: 1-D Winograd F(4,3) on
conv3 / conv4 (reference src/contact_cnn.py:29-43; 60 % of the stack's MFMAs) instead of the F(2,3) that ships (csrc/conv_wino.hip).

numpy only, runs on the CPU.  Both algorithms are emulated in the KERNEL's arithmetic: weights transformed on the host in fp64 and rounded to fp32 once;
input transform in fp32; per Winograd component one fp32 fma chain over the input channels in order (what a v_mfma_f32_16x16x4_f32 accumulator does:
exact product, one rounding per step); output transform in fp32; bias, ReLU, pool.  conv1 / conv2 run on F(2,3) in both arms, the FC layers in fp32 in both
arms, so the two arms differ in conv3 / conv4 only.  Reported: err / bound of the logits against the fp64-accumulating oracle, bound = 1e-5 max|ref| + 1e-4 |ref|
(the contract of every parity test), and of the features (the conv stack's own output) -- over N(0,1) windows, AR(1) windows with offsets (both z-scored),
and pre-normalised windows with a 1e3 dynamic range between channels.

    python tools/wino_f43_emulation.py [windows per set, default 192] > profiles/r6_wino_f43_emulation.json

Stop rule of the review: build a conv4 prototype only if the worst err / bound of the F(4,3) arm stays <= 0.3 AND its 6-point tiles fill 16-column MFMA tiles
with <= 10 % padding."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import synth        # noqa: E402
from oracle import oracle as orc                    # noqa: E402

F32 = np.float32

# F(2,3): y = A^T [(G g) * (B^T d)], tile = 2 outputs from 4 inputs
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)
# F(4,3) (Lavin & Gray, interpolation points 0, +-1, +-2, inf): tile = 4 outputs from 6 inputs
BT4 = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
G4 = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
AT4 = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)


def fma_chain(U, V):
    """acc[n, co, tile, xi] = fp32 fma chain over ci of U[co, ci, xi] * V[n, ci, tile, xi] (exact product, one rounding per step)."""
    n, ci, tiles, nx = V.shape
    acc = np.zeros((n, U.shape[0], tiles, nx), F32)
    for c in range(ci):
        acc = (acc.astype(np.float64) + U[None, :, c, None, :].astype(np.float64) * V[:, None, c, :, :].astype(np.float64)).astype(F32)
    return acc


def f32_matvec(M, x):
    """fp32 evaluation of the small transform M (entries exact in fp32 or rounded once) on the last axis of x, left to right."""
    out = []
    for row in M:
        acc = None
        for coef, col in zip(row, range(x.shape[-1])):
            if coef == 0:
                continue
            term = (F32(coef) * x[..., col]).astype(F32)
            acc = term if acc is None else (acc + term).astype(F32)
        out.append(acc if acc is not None else np.zeros(x.shape[:-1], F32))
    return np.stack(out, -1)


def conv_wino(x, w, b, m):
    """x (n, ci, T) fp32, w (co, ci, 3), zero pad 1: Winograd F(m,3), m in (2, 4), in the kernel's arithmetic; bias + ReLU."""
    BT, G, AT = (BT2, G2, AT2) if m == 2 else (BT4, G4, AT4)
    n, ci, T = x.shape
    tiles = -(-T // m)
    xp = np.zeros((n, ci, tiles * m + 2), F32)
    xp[:, :, 1:T + 1] = x
    d = np.stack([xp[:, :, m * i:m * i + m + 2] for i in range(tiles)], 2)            # (n, ci, tiles, m + 2)
    V = f32_matvec(BT, d)
    U = np.einsum("xk,oik->oix", G, w.astype(np.float64)).astype(F32)                    # host, fp64 -> fp32 once
    M = fma_chain(U, V)
    y = f32_matvec(AT, M)                                                                # (n, co, tiles, m)
    y = y.reshape(n, w.shape[0], tiles * m)[:, :, :T]
    y = (y + b[None, :, None].astype(F32)).astype(F32)      # (the kernel starts component 1's chain from the bias: one rounding earlier, the same order of magnitude)
    return np.maximum(y, F32(0))


def pool(x):
    T = x.shape[2] // 2
    return np.maximum(x[:, :, 0:2 * T:2], x[:, :, 1:2 * T:2])


def forward(sd, zw, m34):
    x = np.ascontiguousarray(zw.transpose(0, 2, 1)).astype(F32)
    x = conv_wino(x, sd["block1.0.weight"], sd["block1.0.bias"], 2)
    x = pool(conv_wino(x, sd["block1.2.weight"], sd["block1.2.bias"], 2))
    x = conv_wino(x, sd["block2.0.weight"], sd["block2.0.bias"], m34)
    x = pool(conv_wino(x, sd["block2.2.weight"], sd["block2.2.bias"], m34))
    feat = x.reshape(x.shape[0], -1)
    h = np.maximum(feat @ sd["fc.0.weight"].T + sd["fc.0.bias"], 0).astype(F32)
    h = np.maximum(h @ sd["fc.3.weight"].T + sd["fc.3.bias"], 0).astype(F32)
    return feat, (h @ sd["fc.6.weight"].T + sd["fc.6.bias"]).astype(F32)


def err_over_bound(got, ref):
    got, ref = got.astype(np.float64), ref.astype(np.float64)
    bound = 1e-5 * np.abs(ref).max() + 1e-4 * np.abs(ref)
    r = np.abs(got - ref) / bound
    return {"max": float(r.max()), "p999": float(np.percentile(r, 99.9)), "mean": float(r.mean())}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    sd = synth.make_state_dict(1, "uniform")
    o = orc.Oracle(sd)
    sets = {}
    sets["normal_zscored"] = orc.zscore_windows(synth.make_sequence(149 + n, seed=3).astype(np.float32))
    sets["ar1_zscored"] = orc.zscore_windows(synth.make_sequence(149 + n, seed=4, kind="ar1").astype(np.float32))
    rng = np.random.default_rng(8)
    wide = rng.standard_normal((n, 150, 54)).astype(np.float32) * (10.0 ** rng.uniform(-1.5, 1.5, (1, 1, 54))).astype(np.float32)
    sets["prenormalised_1e3_channel_range"] = wide
    res = {"windows_per_set": n, "arms": "conv1/conv2 on F(2,3) in both; conv3/conv4 on F(2,3) (what ships) or F(4,3)", "sets": {}}
    for name, zw in sets.items():
        ref = o.forward_windows(zw)
        taps = o.forward_taps(zw) if hasattr(o, "forward_taps") else None
        row = {}
        for m in (2, 4):
            feat, lg = forward(sd, zw, m)
            row[f"F({m},3)"] = {"logits_err_over_bound": err_over_bound(lg, ref["logits"]),
                                "argmax_differences": int((lg.argmax(1) != ref["pred"]).sum())}
            if taps is not None and "feat" in taps:
                row[f"F({m},3)"]["features_err_over_bound"] = err_over_bound(feat, taps["feat"])
        row["ratio_max"] = row["F(4,3)"]["logits_err_over_bound"]["max"] / row["F(2,3)"]["logits_err_over_bound"]["max"]
        res["sets"][name] = row
        print(name, json.dumps(row), file=sys.stderr, flush=True)
    worst = max(r["F(4,3)"]["logits_err_over_bound"]["max"] for r in res["sets"].values())
    # tiling: a stage-2 window is 75 positions = 19 quads (F(4,3)) / 38 pairs (F(2,3)); MFMA column tiles are 16 wide
    tiling = {}
    for wins in (1, 2, 4):
        q, p = 19 * wins, (38 + 1) * wins if wins > 1 else 38      # (F(2,3), two windows: one dummy column per window keeps the column map linear, conv_wino_dev.h ColMap)
        tiling[f"{wins}_windows"] = {"F(4,3)_quads": q, "column_tiles": -(-q // 16), "padding": 1 - q / (16 * -(-q // 16)),
                                     "F(2,3)_pairs": p, "F(2,3)_column_tiles": -(-p // 16), "F(2,3)_padding": 1 - 38 * wins / (16 * -(-p // 16)),
                                     "accumulator_VGPRs_per_wave_at_2_row_tiles": {"F(4,3)": 2 * -(-q // 16) * 6 * 4, "F(2,3)": 2 * -(-p // 16) * 4 * 4}}
    res["tiling_stage2"] = tiling
    res["mfma_per_output"] = {"F(2,3)": 4 / 2, "F(4,3)": 6 / 4, "conv3_conv4_share_of_the_stack's_MFMAs": 0.60,
                              "ideal_conv_stack_time_ratio": 1 - 0.60 * (1 - (6 / 4) / (4 / 2))}
    res["worst_F(4,3)_logits_err_over_bound"] = worst
    res["stop_rule"] = {"worst_err_over_bound_le_0.3": worst <= 0.3,
                        "padding_le_10pct_at_the_shipped_two_windows_per_workgroup": tiling["2_windows"]["padding"] <= 0.10,
                        "accumulators_fit_256_VGPRs_at_a_padding_le_10pct": tiling["4_windows"]["accumulator_VGPRs_per_wave_at_2_row_tiles"]["F(4,3)"] + 40 <= 256}
    res["decision"] = ("prototype conv4" if all(res["stop_rule"].values()) else
                       "closed: " + ", ".join(k for k, v in res["stop_rule"].items() if not v) + " fail(s)")
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Dev check of conv_x3p.hip (two windows per workgroup, a phase apart) against conv_x3.hip (x3_pair=0) and the oracle,
then event-timed A/B of the two; run on the GPU box.  Usage: python tools/dev_x3p.py [--no-oracle] [--time]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from deep_contact_estimator_amd import contact_cnn, synth


def make(precision, pair, max_batch=8192):
    m = contact_cnn(device=0, max_batch=max_batch, precision=precision, tune={"x3_pair": int(pair)})
    m.load_state_dict(synth.make_state_dict(1, "uniform")).eval()
    return m


def main():
    ok = True
    orc = None
    if "--no-oracle" not in sys.argv:
        from oracle import oracle as orc_mod
        orc = orc_mod.Oracle(synth.make_state_dict(1, "uniform"))
    for precision in ("fp32_split", "bf16_fc"):
        a, b = make(precision, True), make(precision, False)
        for n in (4096, 4097, 1024, 5001, 8192):
            x = torch.randn((n, 150, 54), device="cuda", generator=torch.Generator(device="cuda").manual_seed(n))
            ra, rb = a.predict(x), b.predict(x)
            pa, pb = a.last_plan(), b.last_plan()
            ra = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in ra.items()}; rb = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in rb.items()}
            d = np.abs(ra["logits"].astype(np.float64) - rb["logits"]).max()
            scale = np.abs(rb["logits"]).max()
            flips = int((ra["pred"] != rb["pred"]).sum())
            nan = int(np.isnan(ra["logits"]).sum())
            lim = 2e-5 * scale if precision == "fp32_split" else 0.1
            good = d < lim and nan == 0
            ok &= good
            print(f"{precision} n={n}: plan {pa[0]} vs {pb[0]}  max|pair - single| {d:.3e} (scale {scale:.2f})  flips {flips}  nan {nan}  {'OK' if good else 'FAIL'}", flush=True)
            if orc is not None and n == 4097:
                ref = orc.forward_windows(x.cpu().numpy())
                e = np.abs(ra["logits"].astype(np.float64) - ref["logits"])
                bound = 1e-5 * np.abs(ref["logits"]).max() + 1e-4 * np.abs(ref["logits"])
                print(f"    vs oracle: max err/bound {np.max(e / bound):.3f}  argmax differences {int((ra['pred'] != ref['pred']).sum())}", flush=True)
                if precision == "fp32_split":
                    ok &= bool(np.max(e / bound) < 1.0)
        # the z-score entry: raw sequence in, windows never materialised
        T = 6000 + 149
        seq = synth.make_sequence(T, seed=3).astype(np.float32)
        sa, sb = a.infer_sequence(seq), b.infer_sequence(seq)
        d = np.abs(sa["logits"].astype(np.float64) - sb["logits"]).max()
        good = d < (2e-5 * np.abs(sb["logits"]).max() if precision == "fp32_split" else 0.1)
        ok &= good
        print(f"{precision} infer_sequence T={T}: plan {a.last_plan()[0]} vs {b.last_plan()[0]}  max|pair - single| {d:.3e}  flips {int((sa['pred'] != sb['pred']).sum())}  {'OK' if good else 'FAIL'}", flush=True)
        # a window with a non-finite sample: class 0, NaN logits, neighbours untouched
        x = torch.randn((2048, 150, 54), device="cuda")
        x[5, 17, 3] = float("nan"); x[1000, 149, 53] = float("inf")
        r = {k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in a.predict(x).items()}
        bad = np.isnan(r["logits"]).all(1)
        good = bool(bad[5] and bad[1000] and bad.sum() == 2 and r["pred"][5] == 0)
        ok &= good
        print(f"{precision} non-finite windows contained: {'OK' if good else 'FAIL'} ({int(bad.sum())} NaN rows)", flush=True)
        if "--time" in sys.argv:
            x = torch.randn((4096, 150, 54), device="cuda")
            for name, m in (("pair", a), ("single", b)):
                m.profile(1)
                for _ in range(20):
                    m.predict_packed(x)
                m.sync(); m.profile_read()
                t0 = time.perf_counter()
                for _ in range(200):
                    m.predict_packed(x)
                m.sync()
                dt = (time.perf_counter() - t0) / 200
                pr = m.profile_read()
                print(f"{precision} {name}: step {dt * 1e6:.1f} us = {4096 / dt / 1e6:.3f} M windows/s;  kernels {pr}", flush=True)
        a.close(); b.close()
    print("ALL OK" if ok else "FAILURES")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

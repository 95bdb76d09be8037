#!/usr/bin/env python3
"""Interleaved A/B of build / environment variants on the bench step (run on the GPU box).

    python tools/ab_bench.py [--rounds R] [--steps K] [--precision fp32|bf16_fc] name[:KEY=VAL[,KEY=VAL...]] ...

Each variant is `bench.py --no-extras --no-cpu-baseline` in its own process with the given
environment (e.g. DCE_LIB=deep_contact_estimator_amd/libdce_pk0.so, DCE_CONV_GRID=0); variants
are interleaved round by round (cdna_hip_programming.md 5.4 rule 24) and the median over rounds of
windows/s and of every kernel's average launch time is printed, then one JSON line.
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("variants", nargs="+")
    args = ap.parse_args()
    variants = []
    for v in args.variants:
        name, _, envs = v.partition(":")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        variants.append((name, env))
    res = {name: [] for name, _ in variants}
    for r in range(args.rounds):
        for name, env in variants:
            e = dict(os.environ, **env)
            p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-extras", "--no-cpu-baseline",
                                "--steps", str(args.steps), "--warmup", "10", "--precision", args.precision],
                               env=e, capture_output=True, text=True, cwd=ROOT)
            line = [l for l in p.stdout.splitlines() if l.startswith("{")]
            if p.returncode != 0 or not line:
                print(f"[{name}] FAILED rc={p.returncode}\n{p.stderr[-800:]}", flush=True)
                continue
            j = json.loads(line[-1])
            res[name].append(j)
            print(f"round {r} {name}: {j['value'] / 1e6:.4f} M/s  " +
                  "  ".join(f"{k} {v['avg_ms'] * 1e3:.1f}us" for k, v in j["kernels"].items()), flush=True)
    summary = {}
    for name, runs in res.items():
        if not runs:
            continue
        summary[name] = {"windows_per_s": statistics.median(j["value"] for j in runs), "rounds": len(runs),
                         "kernels_us": {k: statistics.median(j["kernels"][k]["avg_ms"] for j in runs) * 1e3
                                        for k in runs[0]["kernels"]}}
    print("\n%-28s %10s  %s" % ("variant", "M win/s", "median kernel us"))
    for name, s in summary.items():
        print("%-28s %10.4f  %s" % (name, s["windows_per_s"] / 1e6, "  ".join(f"{k} {v:.1f}" for k, v in s["kernels_us"].items())))
    print(json.dumps(summary))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn the raw outputs of tools/profile_gpu.sh <tag> (under gpurun_out/) into the tracked
summaries under profiles/: <tag>_kernel_stats.csv, <tag>_pmc_summary.md, pmc_latest.json.
Usage: tools/make_profile_summary.py <tag>"""
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pmc_summary  # noqa: E402

B = 4096
# algorithmic bytes per window (DESIGN.md 4.x / SURVEY.md 8(d)); weights once per launch
ALGO = {
    "conv_stack": 51344 * B + 4 * (96768 + 384),
    "fc1_gemm(128x128)": (18944 + 8192) * B + 4 * (2048 * 4736 + 2048),
    # fc.3 with fc.6's chunk sums in its epilogue: h1 in, 8 x 16 chunk sums out (h2 stays on chip), W2 + W3 once
    "fc2_gemm(64x64)": (8192 + 8 * 64) * B + 4 * (512 * 2048 + 512 + 16 * 512),
    # combine: the chunk sums in, logits + pred + contacts out
    "fc3_tail": (8 * 64 + 64 + 4 + 4) * B + 4 * 16,
}
NAMES = {"conv_stack": "conv_stack", "fc1_gemm(128x128)": "fc1_gemm", "fc2_gemm(64x64)": "fc2_gemm", "fc3_tail": "fc3_tail"}


def main(tag):
    out = os.path.join(ROOT, "gpurun_out")
    shutil.copy(os.path.join(out, f"{tag}_stats", f"{tag}_kernel_stats.csv"), os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"))
    avg_us = {}
    for r in csv.DictReader(open(os.path.join(out, f"{tag}_stats", f"{tag}_kernel_stats.csv"))):
        avg_us[pmc_summary.short(r["Name"])] = float(r["AverageNs"]) / 1e3
    ctr = {}
    for part in ("fetch", "write", "sq", "lds", "tcc1", "tcc2"):
        p = os.path.join(out, f"{tag}_pmc_{part}", f"{tag}_counter_collection.csv")
        if not os.path.exists(p):
            continue
        for k, v in pmc_summary.main([p]).items():
            ctr.setdefault(k, {}).update(v)
    lines = [f"rocprofv3 passes of tools/profile_gpu.sh {tag}; {B} windows per launch. Kernel-trace averages: `python bench.py --steps 300 --warmup 50 "
             "--no-cpu-baseline` (long enough for the clocks to settle: they agree with the HIP-event times in the bench json of the same tag to <1 %); "
             "PMC passes: `--steps 20 --warmup 3`, one counter group per pass.",
             "",
             "| kernel | avg us (kernel-trace) | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes/launch (2*FETCH+WRITE)*1024 | algorithmic bytes/launch | ratio | MFMA busy (SQ_VALU_MFMA_BUSY/1024 / GRBM_GUI_ACTIVE/8) | LDS bank-conflict cycles / LDS active cycles |",
             "|---|---|---|---|---|---|---|---|---|"]
    latest = {}
    for k, algo in ALGO.items():
        c = ctr.get(k)
        if not c:
            continue
        hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c["GRBM_GUI_ACTIVE"] / 8)
        lines.append(f"| {NAMES[k]} | {avg_us.get(k, float('nan')):.1f} | {c['FETCH_SIZE']:.0f} | {c['WRITE_SIZE']:.0f} | {hbm / 1e6:.1f} MB | "
                     f"{algo / 1e6:.1f} MB | {hbm / algo:.2f} | {busy:.3f} | {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} / {c.get('SQ_LDS_IDX_ACTIVE', 0):.3g} |")
        latest[NAMES[k]] = {"hbm_bytes_per_launch": hbm, "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
                            "algorithmic_bytes_per_launch": algo, "mfma_busy_frac": busy, "profile": tag,
                            "note": "FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section (gfx950 counts 128-B requests as 64 B)"}
    # L2 view (optional passes): hit rate and what L2 asks of the fabric.  TCC_EA0_RDREQ counts requests, _32B / _64B the
    # short ones, the rest are 128-byte requests.  Whether a fabric read is served by the memory-side Infinity Cache (MALL)
    # or by HBM is NOT visible to these counters (TCC_EA0_RDREQ_DRAM counts reads addressed to DRAM space, cached or not).
    if any("TCC_HIT" in c for c in ctr.values()):
        lines += ["", "| kernel | L2 hit rate TCC_HIT / (TCC_HIT + TCC_MISS) | TCC_REQ | fabric read requests TCC_EA0_RDREQ | of them 32 B / 64 B | "
                      "fabric read bytes (32/64/128-B requests) | algorithmic bytes / launch |", "|---|---|---|---|---|---|---|"]
        for k, algo in ALGO.items():
            c = ctr.get(k)
            if not c or "TCC_HIT" not in c:
                continue
            hit = c["TCC_HIT"] / max(c["TCC_HIT"] + c["TCC_MISS"], 1)
            rd, r32, r64 = c.get("TCC_EA0_RDREQ", 0), c.get("TCC_EA0_RDREQ_32B", 0), c.get("TCC_EA0_RDREQ_64B", 0)
            rbytes = 32 * r32 + 64 * r64 + 128 * max(rd - r32 - r64, 0)
            lines.append(f"| {NAMES[k]} | {hit:.3f} | {c.get('TCC_REQ', 0):.3g} | {rd:.3g} | {r32:.3g} / {r64:.3g} | {rbytes / 1e6:.1f} MB | {algo / 1e6:.1f} MB |")
            latest[NAMES[k]].update({"l2_hit_rate": hit, "fabric_read_bytes_per_launch": rbytes})
    # bf16-FC mode: per-CU-cycle rates of the GEMMs (GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ counters over all 256 CUs x 4 SIMDs
    # as the fp32 table's MFMA-busy normalisation has it)
    b16 = {}
    for part in ("bf16sq", "bf16lds", "bf16tcc", "bf16fetch", "bf16write"):
        p = os.path.join(out, f"{tag}_pmc_{part}", f"{tag}_counter_collection.csv")
        if os.path.exists(p):
            for k, v in pmc_summary.main([p]).items():
                b16.setdefault(k, {}).update(v)
    if b16:
        lines += ["", "bf16-FC mode (`--precision bf16_fc`), same step:", "",
                  "| kernel | MFMA busy | LDS busy: SQ_LDS_IDX_ACTIVE / (256 CUs x GRBM_GUI_ACTIVE/8) | bank-conflict cycles / LDS active | LDS instructions per launch | L2 hit rate | TCC_REQ |",
                  "|---|---|---|---|---|---|---|"]
        for k in ("conv_stack", "conv_h2", "conv_x3", "fc1_gemm_bf16", "fc2_gemm_bf16"):      # (conv_h2: the mode's conv stack since the end of round 5; conv_x3: before, and with bf16_conv_h2=0)
            c = b16.get(k)
            if not c or "GRBM_GUI_ACTIVE" not in c:
                continue
            cyc = c["GRBM_GUI_ACTIVE"] / 8
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / cyc
            lds = c.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / cyc
            hit = c.get("TCC_HIT", 0) / max(c.get("TCC_HIT", 0) + c.get("TCC_MISS", 0), 1)
            lines.append(f"| {k} | {busy:.3f} | {lds:.3f} | {c.get('SQ_LDS_BANK_CONFLICT', 0):.3g} / {c.get('SQ_LDS_IDX_ACTIVE', 0):.3g} | "
                         f"{c.get('SQ_INSTS_LDS', 0):.3g} | {hit:.3f} | {c.get('TCC_REQ', 0):.3g} |")
    # HBM traffic of the bf16-FC step's kernels against their algorithmic bytes (windows in + bf16 features out; bf16 features + bf16 W1 in, bf16 h1
    # out; bf16 h1 + bf16 W2 + W3 in, chunk sums out)
    algo16 = {"conv_h2": (150 * 54 * 4 + 4736 * 2) * B + 2 * (96768 * 2) + 4 * 384, "conv_x3": (150 * 54 * 4 + 4736 * 2) * B + 2 * (96768 * 3) + 4 * 384, "fc1_gemm_bf16": (4736 * 2 + 2048 * 2) * B + 2 * 2048 * 4736 + 4 * 2048,
              "fc2_gemm_bf16": (2048 * 2 + 8 * 64) * B + 2 * 512 * 2048 + 4 * (512 + 16 * 512)}
    if any("FETCH_SIZE" in c for c in b16.values()):
        lines += ["", "| kernel (bf16-FC step) | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes/launch (2*FETCH+WRITE)*1024 | algorithmic bytes/launch | ratio |", "|---|---|---|---|---|---|"]
        for k, algo in algo16.items():
            c = b16.get(k)
            if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            lines.append(f"| {k} | {c['FETCH_SIZE']:.0f} | {c['WRITE_SIZE']:.0f} | {hbm / 1e6:.1f} MB | {algo / 1e6:.1f} MB | {hbm / algo:.2f} |")
            latest[k + "@bf16_fc"] = {"hbm_bytes_per_launch": hbm, "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "algorithmic_bytes_per_launch": algo, "profile": tag}
    # fp32_f16x2 (tools/profile_f16x2.sh <tag>_f16x2, when it was run on the same build): HBM traffic of the two-term fp16 step's kernels against their
    # algorithmic bytes (windows in + two-term features and scales out; two-term features and W1 in, two-term h1 and scales out; two-term h1 and W2 + W3 in, chunk sums out)
    h2 = {}
    for part in ("fetch", "write", "sq"):
        p = os.path.join(out, f"{tag}_f16x2_pmc_{part}", f"{tag}_f16x2_counter_collection.csv")
        if os.path.exists(p):
            for k, v in pmc_summary.main([p]).items():
                h2.setdefault(k, {}).update(v)
    hp2 = os.path.join(out, f"{tag}_f16x2_source_hash.txt")
    same_build = os.path.exists(hp2) and os.path.exists(os.path.join(out, f"{tag}_source_hash.txt")) and open(hp2).read().strip() == open(os.path.join(out, f"{tag}_source_hash.txt")).read().strip()
    algo_h2 = {"conv_h2": ("conv_stack", (150 * 54 * 4 + 4736 * 4 + 4) * B + 2 * (96768 * 2) + 4 * 384),
               "fc1_gemm_h2": ("fc1_gemm", (4736 * 4 + 2048 * 4 + 4) * B + 4 * 2048 * 4736 + 4 * 2048),
               "fc1_gemm_h2k": ("fc2_gemm", (2048 * 4 + 8 * 64) * B + 4 * 512 * 2048 + 4 * (512 + 16 * 512))}
    if same_build and any("FETCH_SIZE" in c for c in h2.values()):
        lines += ["", "fp32_f16x2 (`--precision fp32_f16x2`), same step, same build:", "",
                  "| kernel | FETCH_SIZE KB | WRITE_SIZE KB | HBM bytes/launch (2*FETCH+WRITE)*1024 | algorithmic bytes/launch | ratio | MFMA busy |", "|---|---|---|---|---|---|---|"]
        for k, (slot, algo) in algo_h2.items():
            c = h2.get(k)
            if not c or "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                continue
            hbm = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024
            busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (c["GRBM_GUI_ACTIVE"] / 8) if "GRBM_GUI_ACTIVE" in c else float("nan")
            lines.append(f"| {k} | {c['FETCH_SIZE']:.0f} | {c['WRITE_SIZE']:.0f} | {hbm / 1e6:.1f} MB | {algo / 1e6:.1f} MB | {hbm / algo:.2f} | {busy:.3f} |")
            latest[slot + "@fp32_f16x2"] = {"hbm_bytes_per_launch": hbm, "fetch_size_kb": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"], "algorithmic_bytes_per_launch": algo,
                                           "mfma_busy_frac": busy, "profile": tag + "_f16x2"}
    # the build these counters were taken on (tools/profile_gpu.sh records the hash of the .so's sources on the box):
    # bench.py marks roofline.traffic stale when the library it times was built from other sources
    hp = os.path.join(out, f"{tag}_source_hash.txt")
    latest["_meta"] = {"profile": tag, "source_hash": open(hp).read().strip() if os.path.exists(hp) else None}
    open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.md"), "w").write("\n".join(lines) + "\n")
    json.dump(latest, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main(sys.argv[1])

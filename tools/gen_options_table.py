#!/usr/bin/env python3
"""DESIGN.md's appendix from the source: the keys of kTuneKeys (csrc/dce_api.hip) with the defaults and comments of struct Tuning
(csrc/dce_kernels.h).    python tools/gen_options_table.py > /tmp/options.md"""
import os, re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
api = open(os.path.join(ROOT, "deep_contact_estimator_amd/csrc/dce_api.hip")).read()
hdr = open(os.path.join(ROOT, "deep_contact_estimator_amd/csrc/dce_kernels.h")).read()
table = api[api.index("const TuneKey kTuneKeys[]"):api.index("#undef TK")]
keys = [(m.group(2), m.group(1) == "TKX") for m in re.finditer(r"(TKX?)\((\w+), '\w'\)", table)]
struct = hdr[hdr.index("struct Tuning {"):hdr.index("// parses \"key=value,...\"")]
info = {}
for line in struct.split("\n"):
    m = re.match(r"\s*(?:bool|int|long long)\s+(.*?);\s*(?://\s*(.*))?$", line)
    if not m:
        continue
    decls, comment = m.group(1), (m.group(2) or "").strip()
    for d in decls.split(","):
        name, _, dflt = d.strip().partition("=")
        info[name.strip()] = (dflt.strip() or "0", comment)
# ---- the plan table (kPlanRows): precision x windows of the launch -> kernel families
rows = api[api.index("constexpr PlanRow kPlanRows[]"):]
rows = rows[:rows.index("};")]
thr = {"T_ONE": "1", "T_CONV16": "`x3_conv_min` (128)", "T_H2FC": "first size with `h2_min_tiles` (96) 256 x 128 tiles of fc.0 (1281)", "T_BF16H2": "`bf16_conv_h2_min` (257)", "T_INF": "-"}
print("| precision | windows from | below | conv stack | fc.0 family | fc.3 family | |\n|---|---|---|---|---|---|---|")
for m in re.finditer(r"\{(DCE_\w+),\s*(T_\w+),\s*(T_\w+),\s*Conv::(\w+),\s*Fam::(\w+),\s*Fam::(\w+),\s*\"([^\"]*)\"\}", rows):
    p, a, b, conv, f0, f3, what = m.groups()
    print(f"| `{p}` | {thr[a]} | {thr[b]} | `{conv}` | {f0} | {f3} | {what} |")
print()
print("| option | default | build | effect |\n|---|---|---|---|")
for k, exp in keys:
    d, c = info.get(k, ("?", ""))
    d = {"true": "1", "false": "0"}.get(d, d)
    print(f"| `{k}` | {d} | {'experiments' if exp else 'product'} | {c.replace('|', '/')} |")

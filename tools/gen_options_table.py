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
print("| option | default | build | effect |\n|---|---|---|---|")
for k, exp in keys:
    d, c = info.get(k, ("?", ""))
    d = {"true": "1", "false": "0"}.get(d, d)
    print(f"| `{k}` | {d} | {'experiments' if exp else 'product'} | {c.replace('|', '/')} |")

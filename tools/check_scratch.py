#!/usr/bin/env python3
"""Every kernel of the default libdce.so: scratch (private segment) bytes, VGPRs, LDS -- from the gfx950 assembly hipcc emits for
each translation unit with the build's own flags (cross-compiles without a GPU).  Exit code 1 if any kernel spills.
    python tools/check_scratch.py [--only=fc_gemm_h2.hip] [-DDCE_EXPERIMENTS=1 ...]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import build
bad, total = [], 0
only = [a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--only=")]
flags = [a for a in sys.argv[1:] if not a.startswith("--only=")]
with tempfile.TemporaryDirectory() as d:
    procs = []
    for s in (only or build.SOURCES):
        out = os.path.join(d, s + ".s")
        procs.append((s, out, subprocess.Popen([build._hipcc(), *build.CFLAGS, *flags, "--cuda-device-only", "-S", os.path.join(build.CSRC, s), "-o", out],
                                                stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)))
    for s, out, p in procs:
        assert p.wait() == 0, s
        txt = open(out).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt):
            total += 1
            name, scratch, vgpr = m.group(1), int(m.group(2)), int(m.group(3))
            if scratch:
                bad.append((s, name, scratch, vgpr))
print(f"{total} kernels in {len(only or build.SOURCES)} translation units; {len(bad)} with scratch")
for b in bad:
    print("  SCRATCH", *b)
sys.exit(1 if bad else 0)

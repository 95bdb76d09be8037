#!/usr/bin/env python3
"""Every kernel of the default libdce.so: scratch (private segment) bytes, VGPRs, LDS -- from the gfx950 assembly hipcc emits for
each translation unit with the build's own flags (cross-compiles without a GPU).  Exit code 1 if any kernel spills.
    python tools/check_scratch.py [--only=fc_gemm_h2.hip] [-DDCE_EXPERIMENTS=1 ...]
Second check on the same assembly (exit code 2): no compiler-generated instruction READS an SCC that an inline-asm statement wrote.  An asm statement that
changes SCC without listing "scc" among its clobbers may be scheduled between an s_add_u32 and its s_addc_u32 -- the carry of a 64-bit address is lost.
That was round 5's intermittent GPU memory fault (fc_gemm_h2.hip: h2_piece).
Third check (exit code 3): the LDS-DMA statements keep their LDS address in m0, which inline asm cannot list as a clobber reliably -- so no
compiler-generated instruction may read or write m0 in a kernel whose asm statements do."""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import build
bad, total, scc_hits, m0_hits = [], 0, [], []
# instructions that read SCC / write SCC (gfx9 SALU; a writer missing from the list can only produce a false alarm, never hide a hit)
SCC_READ = re.compile(r"^\s*(s_addc_u32|s_subb_u32|s_cselect_b32|s_cselect_b64|s_cmov_b32|s_cmov_b64|s_cbranch_scc[01])\b")
SCC_WRITE = re.compile(r"^\s*(s_add_[ui]32|s_sub_[ui]32|s_addc_u32|s_subb_u32|s_addk_i32|s_cmpk?_\w+|s_bitcmp[01]_\w+|s_(and|or|xor|andn2|orn2|nand|nor|xnor)_b(32|64)|s_lsh[lr]_b(32|64)|"
                       r"s_ashr_i(32|64)|s_bfe_[ui](32|64)|s_bfm_dummy|s_(min|max)_[ui]32|s_not_b(32|64)|s_abs_i32|s_absdiff_i32|s_lshl[1-4]_add_u32|s_wqm_b(32|64)|s_quadmask_b(32|64)|"
                       r"s_bcnt[01]_i32_b(32|64)|s_\w+_saveexec_b64|s_andn[12]_wrexec_b64)\b")


def scan_scc(src, txt):
    kern, tainted, in_asm = None, None, False
    asm_m0, cc_m0 = {}, {}
    for ln, l in enumerate(txt.splitlines(), 1):
        if re.search(r"\bm0\b", l) and not l.lstrip().startswith((";", ".")):
            (asm_m0 if in_asm else cc_m0).setdefault(kern, []).append((ln, l.strip()))
        m = re.match(r"^(_Z\w+):", l)
        if m:
            kern, tainted = m.group(1), None
        elif "#ASMSTART" in l:
            in_asm = True
        elif "#ASMEND" in l:
            in_asm = False
        elif not in_asm and SCC_READ.match(l) and tainted:
            scc_hits.append((src, kern, tainted, ln, l.strip()))
        if SCC_WRITE.match(l):
            tainted = (ln, l.strip()) if in_asm else None
    for k in asm_m0:
        if k in cc_m0:
            m0_hits.append((src, k, cc_m0[k][0]))
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
        scan_scc(s, txt)
        for m in re.finditer(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)", txt):
            total += 1
            name, scratch, vgpr = m.group(1), int(m.group(2)), int(m.group(3))
            if scratch:
                bad.append((s, name, scratch, vgpr))
print(f"{total} kernels in {len(only or build.SOURCES)} translation units; {len(bad)} with scratch")
for b in bad:
    print("  SCRATCH", *b)
print(f"{len(scc_hits)} reads of an SCC written inside an inline-asm statement")
for h in scc_hits[:12]:
    print("  SCC", h[0], h[1][:100], "asm line", h[2], "-> line", h[3], h[4])
print(f"{len(m0_hits)} kernels in which compiler-generated code touches m0 beside asm statements that own it")
for h in m0_hits[:8]:
    print("  M0", h[0], h[1][:100], h[2])
sys.exit(1 if bad else 2 if scc_hits else 3 if m0_hits else 0)

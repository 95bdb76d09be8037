"""CPU: what the build must hold for the memory-safety argument of DESIGN.md 4.6, read off the gfx950 assembly hipcc emits for every translation unit.

1. No compiler-generated instruction reads an SCC that an inline-asm statement wrote.  Round 5's intermittent GPU memory fault was exactly that: the
   statement that issues an LDS-DMA piece (fc_gemm_h2.hip: h2_piece) bumps m0 with s_add_u32, did not list "scc" among its clobbers, and was scheduled
   between the s_add_u32 and the s_addc_u32 of the next piece's 64-bit base -- the carry lost, rows fetched from 4 GB below whenever the operand
   buffer crosses a multiple of 4 GB (profiles/r6l_scc_clobber_ab.txt).  The scan finds the old statement (22 sites, the shipped fc.3 kernel among
   them) and nothing in today's builds.
2. NO kernel of the product library uses scratch (private-segment) memory; the experiments build has exactly one that does (the fc.0 K-split variant,
   27 spilled VGPRs).  Scratch was the first suspect of that fault and was cleared (profiles/r6k_ksplit_nospill_ab.txt); the property stays checked."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _check(*args):
    return subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_scratch.py"), *args], capture_output=True, text=True, timeout=900, cwd=ROOT)


def test_no_product_kernel_uses_scratch():
    r = _check()
    assert r.returncode == 0 and " 0 with scratch" in r.stdout, r.stdout + r.stderr
    n = int(r.stdout.split()[0])
    assert n >= 60, r.stdout                                   # every translation unit was really compiled and parsed


def test_the_one_scratch_kernel_is_the_experiments_only_fc0_k_split():
    r = _check("--only=fc_gemm_h2.hip", "-DDCE_EXPERIMENTS=1")
    lines = [l for l in r.stdout.splitlines() if "SCRATCH" in l]
    assert r.returncode == 1 and len(lines) == 1, r.stdout + r.stderr
    assert "fc_gemm_h2k_kernel" in lines[0] and "H2KCfg" in lines[0] and "256" in lines[0], lines[0]


def test_no_compiler_instruction_reads_an_scc_written_by_inline_asm():
    r = _check()
    assert r.returncode == 0 and "\n0 reads of an SCC written inside an inline-asm statement" in r.stdout, r.stdout + r.stderr
    assert "\n0 kernels in which compiler-generated code touches m0" in r.stdout, r.stdout       # (the LDS-DMA statements own m0 where they use it)
    r = _check("--only=fc_gemm_h2.hip", "--only=fc_gemm_x3.hip", "--only=fc_gemm_phased.hip", "--only=conv_x3p.hip", "-DDCE_EXPERIMENTS=1")
    assert "\n0 reads of an SCC written inside an inline-asm statement" in r.stdout, r.stdout + r.stderr
    assert "\n0 kernels in which compiler-generated code touches m0" in r.stdout, r.stdout


def test_the_scan_sees_the_statement_that_caused_round_5s_fault():
    """the old asm statement (-DH2_SCC_UNDECLARED=1, experiments builds only): the s_addc_u32 of the next piece's base reads the asm's SCC, in all three
    K-split instantiations -- the one that ships as fp32_f16x2's fc.3 among them"""
    r = _check("--only=fc_gemm_h2.hip", "-DDCE_EXPERIMENTS=1", "-DH2_SCC_UNDECLARED=1")
    hits = [l for l in r.stdout.splitlines() if l.lstrip().startswith("SCC ")]
    assert r.returncode != 0 and hits and all("s_addc_u32" in l and "fc_gemm_h2k_kernel" in l for l in hits), r.stdout + r.stderr
    n = int([l for l in r.stdout.splitlines() if "reads of an SCC" in l][0].split()[0])
    assert n >= 10, r.stdout
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from deep_contact_estimator_amd import build; import subprocess, os; "
                        "sys.exit(subprocess.run([build._hipcc(), *build.CFLAGS, '-DH2_SCC_UNDECLARED=1', '--cuda-device-only', '-S', os.path.join(build.CSRC, 'fc_gemm_h2.hip'), '-o', os.devnull], "
                        "capture_output=True).returncode)" % ROOT], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0                                     # the product build refuses the macro (#error)

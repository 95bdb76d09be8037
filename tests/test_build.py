"""CPU: what the build must hold for the memory-safety argument of DESIGN.md 4.6 -- NO kernel of the product library uses scratch (private-segment)
memory.  Round 5's intermittent GPU memory fault sat in the one kernel of either build that does: the experiments-only fc.0 instantiation of
fc_gemm_h2k_kernel (27 spilled VGPRs; the faulting addresses lie in the private-segment aperture, profiles/r6d_ksplit_fault_trace.txt).  Its sibling
instantiations -- what ships on fp32_f16x2's fc.3 -- spill nothing, and this test keeps it that way for every kernel hipcc emits for gfx950."""
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

"""GPU (-m gpu), round 6: memory safety of the product path under GUARD PLACEMENT (csrc/dev_alloc.hip, option guard_alloc): every device buffer of the
context -- and, with --device-io, the caller's -- sits in a mapping of its own whose last (1) / first (2) byte abuts an unmapped page, so a kernel
access one element outside ANY buffer is a GPU page fault at that access, whatever the allocator's history; placement 3 puts a multiple of 4 GB INSIDE
every buffer, with the rest of the reservation unmapped.  tools/guard_stress.py runs create / run {1281, 3072, 4100, 8192, 12289 windows + a 700-window
raw sequence} / destroy cycles in a process of its own (a fault kills the process) and compares every cycle with a plain-allocation context bit for bit.

Why: round 5 saw an intermittent "Memory access fault by GPU" in an experiments-build instantiation of fc_gemm_h2k_kernel (fc.0 on 64 x 128 wave
tiles) whose sibling instantiation ships as fp32_f16x2's fc.3.  Round 6 found it (profiles/r6d, r6k, r6l): the inline-asm statement that
issues an LDS-DMA piece bumps m0 with s_add_u32 -- which rewrites SCC -- and did not list "scc" among its clobbers; the scheduler put it between the
s_add_u32 and the s_addc_u32 that form the next piece's 64-bit base, the carry was lost, and a piece whose rows lie beyond a multiple of 4 GB that
the operand buffer happens to cross was fetched from 4 GB below (unmapped -> fault; mapped -> silently wrong rows).  All three K-split instantiations
had it in their assembly, the shipped one among them.  tests/test_build.py scans the build's assembly for the pattern; placement 3 here is the
run-time check that no kernel of the product forms an address that way."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stress(*args, expect_fault=False, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_stress.py"), *args], env=dict(os.environ, PYTHONPATH=ROOT),
                       capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    if expect_fault:
        assert r.returncode != 0 and "Memory access fault" in (r.stdout + r.stderr), (r.returncode, (r.stdout + r.stderr)[-1500:])
        return None
    assert r.returncode == 0, (r.returncode, (r.stdout + r.stderr)[-3000:])
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["ok"] and not j["mismatches"], j
    return j


def test_guard_placement_catches_an_overrun_of_one_window():
    """The instrument works: a caller that claims ONE window more than its guard-placed input buffer holds dies with a GPU memory fault."""
    _stress("--precision", "fp32", "--cycles", "1", "--guard", "1", "--device-io", "--overrun", "1", "--sizes", "1281", "--sequence", "0", expect_fault=True)


@pytest.mark.parametrize("precision", ["fp32", "fp32_f16x2", "bf16_fc"])
def test_product_precisions_stay_inside_their_buffers(precision):
    """200 create / run / destroy cycles with tail-abutting buffers, the caller's included (device pointers, as bench.py and torch callers pass them);
    40 with host pointers (the context's staging ring); 40 with HEAD-abutting buffers.  No fault, every cycle bit-identical to a plain context."""
    j = _stress("--precision", precision, "--cycles", "200", "--guard", "1", "--device-io")
    assert j["cycles"] == 200 and j["sizes"] == [1281, 3072, 4100, 8192, 12289]
    if precision == "fp32_f16x2":                              # (the sibling of round 5's faulting instantiation is what these launches run)
        assert "fc23_fused_h2_128x64" in j["plans"]["win4100"] and "fc_h2_256x128_out2" in j["plans"]["win12289"], j["plans"]
    _stress("--precision", precision, "--cycles", "40", "--guard", "1")
    _stress("--precision", precision, "--cycles", "40", "--guard", "2", "--device-io")


@pytest.mark.parametrize("precision", ["fp32", "fp32_f16x2", "bf16_fc"])
def test_product_precisions_with_a_4gb_line_inside_every_buffer(precision):
    """Placement 3: a multiple of 4 GB inside every buffer of the context and of the caller -- a 64-bit address whose carry was lost lands on an unmapped
    page.  (The statement that lost it faults here in the first cycle: tools/guard_trace7.sh, profiles/r6l_scc_clobber_ab.txt.)"""
    j = _stress("--precision", precision, "--cycles", "12", "--guard", "3", "--device-io")
    assert j["guard"] == 3 and j["cycles"] == 12


def test_the_undeclared_scc_statement_faults_under_placement_3():
    """The A/B build with the OLD asm statement (libdce_h2sccbug.so, when it has been built: tools/guard_trace7.sh says how) dies in its first guarded
    cycle on the PRODUCT plan of fp32_f16x2 -- the instrument sees the bug the fix removed."""
    lib = os.path.join(ROOT, "deep_contact_estimator_amd", "libdce_h2sccbug.so")
    if not os.path.exists(lib):
        pytest.skip("libdce_h2sccbug.so not built (build.build_variant('h2sccbug', ['-DDCE_EXPERIMENTS=1', '-DH2_SCC_UNDECLARED=1']))")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "guard_stress.py"), "--precision", "fp32_f16x2", "--cycles", "2", "--guard", "3", "--device-io",
                        "--sizes", "4100", "--sequence", "0"], env=dict(os.environ, PYTHONPATH=ROOT, DCE_LIB=lib), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "Memory access fault" in (r.stdout + r.stderr), (r.returncode, (r.stdout + r.stderr)[-1500:])


def test_latency_mode_stays_inside_its_buffers():
    """The latency mode's kernels (one window, 2 .. 32 windows) with guard-placed exchange buffers: tools/guard_stress.py on small sizes."""
    _stress("--precision", "fp32", "--cycles", "60", "--guard", "1", "--device-io", "--tune", "latency=1", "--sizes", "1,2,16,17,30,32,33", "--sequence", "20", "--max-batch", "64")
    _stress("--precision", "fp32", "--cycles", "20", "--guard", "2", "--device-io", "--tune", "latency=1", "--sizes", "1,2,16,17,30,32,33", "--sequence", "20", "--max-batch", "64")
    _stress("--precision", "fp32", "--cycles", "10", "--guard", "3", "--device-io", "--tune", "latency=1", "--sizes", "1,2,16,17,30,32,33", "--sequence", "20", "--max-batch", "64")

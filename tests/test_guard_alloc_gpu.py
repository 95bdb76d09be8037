"""GPU (-m gpu), round 6: memory safety of the product path under GUARD PLACEMENT (csrc/dev_alloc.hip, option guard_alloc): every device buffer of the
context -- and, with --device-io, the caller's -- sits in a mapping of its own whose last (1) / first (2) byte abuts an unmapped page, so a kernel
access one element outside ANY buffer is a GPU page fault at that access, whatever the allocator's history.  tools/guard_stress.py runs create /
run {1281, 3072, 4100, 8192, 12289 windows + a 700-window raw sequence} / destroy cycles in a process of its own (a fault kills the process) and
compares every cycle with a plain-allocation context bit for bit.

Why: round 5 saw an intermittent "Memory access fault by GPU" in an experiments-build instantiation of fc_gemm_h2k_kernel (fc.0 on 64 x 128 wave
tiles) whose sibling instantiation ships as fp32_f16x2's fc.3.  Round 6 traced it (profiles/r6d_ksplit_fault_trace.txt): the faulting
addresses lie in the PRIVATE-SEGMENT aperture -- a 4 GB-aligned base plus a wave's scratch offset, ~4 GB away from every buffer of the context --
i.e. they are accesses to the 27 spilled VGPRs of that one instantiation (the only kernel of either build with scratch), not to an operand.  The
product library holds no kernel with scratch (tests/test_build.py checks the build), and this file checks what remains to be checked: that no
product kernel reads or writes outside its operands."""
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


def test_latency_mode_stays_inside_its_buffers():
    """The latency mode's kernels (one window, 2 .. 32 windows) with guard-placed exchange buffers: tools/guard_stress.py on small sizes."""
    _stress("--precision", "fp32", "--cycles", "60", "--guard", "1", "--device-io", "--tune", "latency=1", "--sizes", "1,2,16,17,30,32,33", "--sequence", "20", "--max-batch", "64")
    _stress("--precision", "fp32", "--cycles", "20", "--guard", "2", "--device-io", "--tune", "latency=1", "--sizes", "1,2,16,17,30,32,33", "--sequence", "20", "--max-batch", "64")

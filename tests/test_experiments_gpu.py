"""GPU (-m gpu): the A/B kernel variants live in a library of their own (libdce_experiments.so, -DDCE_EXPERIMENTS=1: the four-row-tile
Winograd workgroup, the lockstep GEMM schedule, the paired three-term conv stack).  The product library carries none of them; the
tests that exercise them (marked `experiments`, skipped in the default run) plus the tests that loop over kernel families /
schedules are re-run here in a subprocess with DCE_LIB pointing at that library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_product_library_has_no_experiment_kernels():
    """(CPU) dce_build_flags() == 0 for libdce.so, and its switches for the variants are inert by construction."""
    from deep_contact_estimator_amd import _lib
    if os.environ.get("DCE_LIB"):
        pytest.skip("DCE_LIB names another build")
    assert _lib.load().dce_build_flags() == 0


@pytest.mark.gpu
def test_experiment_variants_in_their_own_build():
    if os.environ.get("DCE_LIB"):
        pytest.skip("already running under a DCE_LIB build")
    from deep_contact_estimator_amd import build
    lib = build.build_experiments()                       # (reused only if its recorded source hash matches the sources)
    env = dict(os.environ, DCE_LIB=lib, PYTHONPATH=ROOT)
    sel = ("rt4 or paired or ab_switches or layer_taps_bit_identical or tapped_kernels or phased_gemm_equals_tile or reference_hooks "
           "or k_tiles or barrier_free or persistent or direct_form or bench_batch or k32 or dealt_out_between_the_wave_groups "
           "or split or x3 or three_term or fp32_split or planes or chip_ar1 or permk")      # ... and, since round 6, everything of DCE_FP32_SPLIT      # every experiments-marked test runs here once
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_round3_gpu.py"), os.path.join(ROOT, "tests", "test_round4_gpu.py"),
                        os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_round5_gpu.py"), os.path.join(ROOT, "tests", "test_f16x2_gpu.py"),
                        os.path.join(ROOT, "tests", "test_x3_gpu.py")], env=env, capture_output=True, text=True, timeout=2400, cwd=ROOT)
    tail = (r.stdout + r.stderr)[-3000:]
    assert r.returncode == 0, tail
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1] and "failed" not in r.stdout.splitlines()[-1], tail
    print(r.stdout.strip().splitlines()[-1])

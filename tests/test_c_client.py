"""The C ABI from plain C: tests/c/abi_client.c is compiled with gcc against include/dce.h and
libdce.so (no Python, no torch in that process) with the CPU oracle linked in as the checker.
CPU: it must compile and link against every symbol it uses.  GPU: it must run green."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "deep_contact_estimator_amd")
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")


def _build_client(out):
    if not os.path.exists(os.path.join(PKG, "libdce.so")):
        from deep_contact_estimator_amd import build
        build.build()
    cmd = ["gcc", "-O2", "-std=c11", "-fopenmp", "-Wall", "-Werror",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "oracle"),
           os.path.join(ROOT, "tests", "c", "abi_client.c"), os.path.join(ROOT, "oracle", "dce_oracle.c"),
           "-L" + PKG, "-ldce", "-L" + ROCM_LIB, "-lamdhip64", "-lm",
           "-Wl,-rpath," + PKG, "-Wl,-rpath," + ROCM_LIB, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_c_client_compiles_and_links(tmp_path):
    exe = _build_client(str(tmp_path / "abi_client"))
    assert os.access(exe, os.X_OK)


@pytest.mark.gpu
def test_c_client_runs(tmp_path):
    exe = _build_client(str(tmp_path / "abi_client"))
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_client: OK" in r.stdout
    print(r.stdout.strip())

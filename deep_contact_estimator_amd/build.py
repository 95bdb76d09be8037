"""Build libdce.so (the HIP kernels + C ABI) in-tree for gfx950.

    python -m deep_contact_estimator_amd.build [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container.  The
.so stays next to the package (git-ignored, but it travels to the GPU box with the snapshot).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libdce.so")
SOURCES = ["conv_stack.hip", "conv_wino.hip", "latency.hip", "latency_mb.hip", "conv_x3.hip", "conv_x3p.hip", "conv_h2.hip", "fc_gemm.hip", "fc_gemm_phased.hip", "fc_gemm_chain.hip", "fc_gemm_split.hip", "fc_gemm_x3.hip", "fc_gemm_h2.hip", "fc_gemv.hip", "fc_stream_bf16.hip", "dce_api.hip", "dce_comm.hip", "dev_alloc.hip"]
HEADERS = ["dce_kernels.h", "dce_ctx.h", "fc_tree.h", "conv_common.h", "conv_wino_dev.h", "conv_x3_common.h", "fc6_chain.h", os.path.join("..", "..", "include", "dce.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
OBJDIR = os.path.join(HERE, "build")


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libdce.so)")


def source_hash() -> str:
    """sha256 over the sources libdce.so is built from (names + contents, fixed order)."""
    import hashlib
    h = hashlib.sha256()
    for rel in sorted(SOURCES + HEADERS):
        path = os.path.normpath(os.path.join(CSRC, rel))
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def built_hash(lib: str = LIB) -> str | None:
    """The source hash recorded next to libdce.so when it was built (it travels with the .so to the GPU box):
    ties measurements (profiles/pmc_latest.json) to the build they were taken on."""
    try:
        return open(lib + ".srchash").read().strip()
    except OSError:
        return None


def stale() -> bool:
    if not os.path.exists(LIB) or built_hash() is None:
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines: list[str], force: bool = True) -> str:
    """Experimental A/B build: libdce_<name>.so with extra -D flags (select with DCE_LIB=...).  The translation units compile in
    parallel; the source hash is recorded next to the library (as for libdce.so) and a library whose record matches the sources
    and the flags is reused unless `force`."""
    out = os.path.join(HERE, f"libdce_{name}.so")
    stamp = source_hash() + " " + " ".join(defines)
    if not force and os.path.exists(out) and built_hash(out) == stamp:
        return out
    hipcc = _hipcc()
    od = os.path.join(OBJDIR, name)
    os.makedirs(od, exist_ok=True)
    procs = []
    for s in SOURCES:
        obj = os.path.join(od, s.replace(".hip", ".o"))
        cmd = [hipcc, *CFLAGS, *defines, "-c", os.path.join(CSRC, s), "-o", obj]
        procs.append((cmd, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for cmd, obj, p in procs:
        log, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + log)
        objs.append(obj)
    subprocess.run([shutil.which("g++") or "g++", "-shared", "-fPIC", *objs, "-o", out + ".tmp"], check=True)
    os.replace(out + ".tmp", out)
    with open(out + ".srchash", "w") as f:
        f.write(stamp + "\n")
    return out


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc -c each translation unit for gfx950, then link WITHOUT naming a HIP runtime:
    libdce.so leaves hip* undefined so that it binds to the ONE runtime already in the process
    (PyTorch wheels bundle their own libamdhip64/libhsa-runtime64; a second copy from
    /opt/rocm in the same process cannot open the device).  _lib.load() puts the right runtime
    in the global symbol scope first."""
    if not force and not stale():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    hipcc = _hipcc()
    procs = []
    for s in SOURCES:
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        cmd = [hipcc, *CFLAGS, "-c", os.path.join(CSRC, s), "-o", obj]
        procs.append((cmd, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs = []
    for cmd, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + out)
        if verbose and out:
            print(out, file=sys.stderr)
        objs.append(obj)
    link = [shutil.which("g++") or "g++", "-shared", "-fPIC", *objs, "-o", LIB + ".tmp"]
    r = subprocess.run(link, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + " ".join(link) + "\n" + r.stdout + r.stderr)
    os.replace(LIB + ".tmp", LIB)
    with open(LIB + ".srchash", "w") as f:
        f.write(source_hash() + "\n")
    return LIB


def build_experiments(force: bool = False) -> str:
    """libdce_experiments.so: the product kernels PLUS the variants that were measured slower and kept for the A/B (the
    four-row-tile Winograd workgroup, the lockstep GEMM schedule, the paired three-term conv stack) and the probe macros.
    Select with DCE_LIB=<path>; dce_build_flags() & DCE_BUILD_EXPERIMENTS tells which one is loaded."""
    return build_variant("experiments", ["-DDCE_EXPERIMENTS=1"], force=force)


def build_asan() -> str:
    """libdce_asan.so: the HOST side of every translation unit (the C ABI shim: staging ring, per-thread tuning scope, the
    RCCL dlopen path) instrumented with AddressSanitizer + UndefinedBehaviorSanitizer; device code unchanged.  Run with
    LD_PRELOAD=$(asan_runtime()) ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 DCE_LIB=<path> (tools/run_asan.sh)."""
    return build_variant("asan", ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g1", "-O1"])


def asan_runtime() -> str:
    out = subprocess.run([_hipcc(), "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    return out


if __name__ == "__main__":
    if "--experiments" in sys.argv:
        print(build_experiments())
    elif "--asan" in sys.argv:
        print(build_asan())
    else:
        print(build(force="--force" in sys.argv, verbose=True))

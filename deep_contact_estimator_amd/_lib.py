"""ctypes binding of libdce.so -- the only way the Python host reaches the GPU.

There is deliberately NO fallback: if the HIP library is missing or no MI355X is visible the
product fails loudly (RuntimeError), it never computes on the CPU.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCE_LIB") or os.path.join(_HERE, "libdce.so")   # DCE_LIB: A/B builds

# every symbol include/dce.h declares: (name, restype, argtypes)
_i64p = C.POINTER(C.c_int64)
SYMBOLS = {
    "dce_abi_version": (C.c_int, []),
    "dce_build_flags": (C.c_int, []),
    "dce_device_count": (C.c_int, []),
    "dce_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64]),
    "dce_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_char_p]),
    "dce_split_guard_info": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dce_destroy": (None, [C.c_void_p]),
    "dce_set_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "dce_load_weight": (C.c_int, [C.c_void_p, C.c_char_p, C.c_void_p, _i64p, C.c_int]),
    "dce_finalize_weights": (C.c_int, [C.c_void_p, C.c_int]),
    "dce_forward_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dce_infer_sequence": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dce_forward_windows_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dce_infer_sequence_packed": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "dce_unpack_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dce_comm_get_unique_id": (C.c_int, [C.c_void_p]),
    "dce_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "dce_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "dce_gather_results": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, _i64p, C.c_int, C.c_int]),
    "dce_allreduce_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "dce_comm_sync": (C.c_int, [C.c_void_p]),
    "dce_comm_destroy": (C.c_int, [C.c_void_p]),
    "dce_zscore_windows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_void_p]),
    "dce_forward_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dce_conv_layer_taps": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int] + [C.c_void_p] * 6),
    "dce_confusion_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "dce_online_reset": (C.c_int, [C.c_void_p]),
    "dce_online_push": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dce_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "dce_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), _i64p, C.c_int]),
    "dce_last_plan": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int]),
    "dce_sync": (C.c_int, [C.c_void_p]),
    "dce_last_error": (C.c_char_p, [C.c_void_p]),
    "dce_debug_split3": (None, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "dce_debug_split_h2": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p]),
    "dce_debug_latency_trace": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dce_debug_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "dce_debug_free": (C.c_int, [C.c_void_p, C.c_void_p]),
}

BUILD_EXPERIMENTS, BUILD_TRACE, BUILD_ASAN = 1, 2, 4


class SplitGuard(C.Structure):
    """include/dce.h dce_split_guard"""
    _fields_ = [("precision", C.c_int), ("enabled", C.c_int), ("refused", C.c_int), ("x_hi", C.c_float), ("x_lo", C.c_float),
                ("z_max", C.c_float), ("gain", C.c_double * 6), ("offs", C.c_double * 6), ("guarded_launches", C.c_uint32),
                ("windows_out_of_range", C.c_uint32), ("fallbacks_run", C.c_uint32), ("reason", C.c_char * 256)]


def tune_spec(tune) -> bytes | None:
    """dict / str -> the option string of dce_create_ex ("key=value,key=value"); None: the library reads DCE_TUNE."""
    if tune is None:
        return None
    if isinstance(tune, str):
        return tune.encode()
    return ",".join(f"{k}={int(v)}" for k, v in tune.items()).encode()


def has_experiments() -> bool:
    """Is the loaded library the experiments build (the A/B kernel variants exist and their switches work)?"""
    return bool(load().dce_build_flags() & BUILD_EXPERIMENTS)

_lib = None


def _load_hip_runtime():
    """Put exactly one HIP runtime into the global symbol scope before libdce.so is opened.

    PyTorch-ROCm wheels bundle their own libamdhip64.so + libhsa-runtime64.so; device pointers
    and streams handed over from torch tensors belong to THAT runtime, and a second runtime
    (e.g. /opt/rocm's) in the same process cannot even open the device.  So: torch's copy when
    torch is installed, the system ROCm one otherwise."""
    cands = []
    try:
        import torch                                  # noqa: F401  (loads its bundled runtime)
        cands.append(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    except Exception:
        pass
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cands += [os.path.join(rocm, "lib", "libamdhip64.so"), "libamdhip64.so.7", "libamdhip64.so"]
    last = None
    for c in cands:
        if os.path.isabs(c) and not os.path.exists(c):
            continue
        try:
            return C.CDLL(c, mode=C.RTLD_GLOBAL)
        except OSError as e:                          # try the next candidate
            last = e
    raise RuntimeError(f"no HIP runtime (libamdhip64) could be loaded: {last}")


def load():
    """dlopen libdce.so and bind every declared symbol.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m deep_contact_estimator_amd.build` "
            "(there is no CPU fallback for the inference path)")
    _load_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)          # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def use_process_rccl():
    """Point libdce.so's run-time RCCL binding (csrc/dce_comm.hip) at the librccl that belongs to the HIP runtime of
    this process: torch's bundled copy when torch is installed (its librccl is linked to its own libamdhip64), the
    system ROCm one otherwise.  DCE_RCCL_LIB set by the user wins."""
    if os.environ.get("DCE_RCCL_LIB"):
        return
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(cand):
            os.environ["DCE_RCCL_LIB"] = cand
    except Exception:
        pass


class DceError(RuntimeError):
    pass


def check(rc: int, ctx=None):
    if rc != 0:
        msg = load().dce_last_error(ctx)
        raise DceError(f"libdce error {rc}: {msg.decode() if msg else '?'}")

"""ctypes binding of oracle/libdce_oracle.so (dce_oracle.c) -- TEST INFRASTRUCTURE ONLY.

Parity status: pinned by tests/test_oracle.py against tests/golden/*.npz (generated from the
imported reference by tests/golden/make_golden.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdce_oracle.so")

_KEYS = (
    ("block1.0.weight", "block1.0.bias"), ("block1.2.weight", "block1.2.bias"),
    ("block2.0.weight", "block2.0.bias"), ("block2.2.weight", "block2.2.bias"),
    ("fc.0.weight", "fc.0.bias"), ("fc.3.weight", "fc.3.bias"), ("fc.6.weight", "fc.6.bias"),
)


class _Weights(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "c1w", "c1b", "c2w", "c2b", "c3w", "c3b", "c4w", "c4b",
        "f1w", "f1b", "f2w", "f2b", "f3w", "f3b")]


def build(force: bool = False) -> str:
    """Compile dce_oracle.c with gcc if the .so is missing or stale."""
    src = os.path.join(_HERE, "dce_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B"], check=True, capture_output=True)
    return _LIB_PATH


_lib = None
_default_threads = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_argmax16.restype = C.c_int32
        _lib.oracle_bf16_round.restype = C.c_float
        _lib.oracle_bf16_round.argtypes = [C.c_float]
    return _lib


def set_threads(n: int) -> None:
    """OpenMP threads of the windows loop (oracle_forward_windows / oracle_infer_sequence):
    n = 1 for the single-thread timing of bench.py's cpu_baseline, n <= 0 restores one per CPU."""
    global _default_threads
    lib()
    gomp = C.CDLL("libgomp.so.1")
    if _default_threads is None:
        _default_threads = gomp.omp_get_max_threads()
    gomp.omp_set_num_threads(int(n) if n > 0 else _default_threads)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Holds a state_dict (numpy fp32, PyTorch layouts) and evaluates the reference path."""

    def __init__(self, state_dict, bf16_fc: bool = False):
        """bf16_fc: BASELINE configs[4] -- fc.0 / fc.3 weights rounded to bf16 here (once), their input activations
        rounded inside the C restatement; everything else fp32."""
        self._keep = []
        self.bf16_fc = bool(bf16_fc)
        if self.bf16_fc:
            state_dict = dict(state_dict)
            for k in ("fc.0.weight", "fc.3.weight"):
                state_dict[k] = bf16_round(state_dict[k])
        w = _Weights()
        names = [f[0] for f in _Weights._fields_]
        i = 0
        for wk, bk in _KEYS:
            for k in (wk, bk):
                a = np.ascontiguousarray(np.asarray(state_dict[k]), dtype=np.float32)
                self._keep.append(a)
                setattr(w, names[i], a.ctypes.data)
                i += 1
        self._w = w

    def forward_windows(self, windows, taps: bool = False):
        """windows (n,150,54) pre-normalised -> dict(logits, pred, contacts[, feat, h1, h2])."""
        x = np.ascontiguousarray(windows, dtype=np.float32)
        n = x.shape[0]
        assert x.shape[1:] == (150, 54)
        out = {
            "logits": np.empty((n, 16), np.float32),
            "pred": np.empty((n,), np.int32),
            "contacts": np.empty((n, 4), np.uint8),
        }
        if taps:
            out["feat"] = np.empty((n, 4736), np.float32)
            out["h1"] = np.empty((n, 2048), np.float32)
            out["h2"] = np.empty((n, 512), np.float32)
        fn = lib().oracle_forward_windows_bf16fc if self.bf16_fc else lib().oracle_forward_windows
        rc = fn(
            C.byref(self._w), _p(x), C.c_int64(n), _p(out.get("feat")), _p(out.get("h1")),
            _p(out.get("h2")), _p(out["logits"]), _p(out["pred"]), _p(out["contacts"]))
        assert rc == 0
        return out

    def infer_sequence(self, seq, want_windows: bool = False):
        """seq (T,54) raw -> dict(logits (N,16), pred (N,), contacts (N,4)[, windows])."""
        s = np.ascontiguousarray(seq, dtype=np.float32)
        T = s.shape[0]
        n = max(T - 149, 0)
        out = {
            "logits": np.empty((n, 16), np.float32),
            "pred": np.empty((n,), np.int32),
            "contacts": np.empty((n, 4), np.uint8),
        }
        if want_windows:
            out["windows"] = np.empty((n, 150, 54), np.float32)
        rc = lib().oracle_infer_sequence(
            C.byref(self._w), _p(s), C.c_int64(T), _p(out.get("windows")),
            _p(out["logits"]), _p(out["pred"]), _p(out["contacts"]))
        assert rc == 0
        return out

    def layer_taps(self, window):
        """One pre-normalised window (150,54) -> per-layer activations in [C][T] layout."""
        x = np.ascontiguousarray(window, dtype=np.float32)
        assert x.shape == (150, 54)
        out = {
            "conv1": np.empty((64, 150), np.float32), "conv2": np.empty((64, 150), np.float32),
            "pool1": np.empty((64, 75), np.float32), "conv3": np.empty((128, 75), np.float32),
            "conv4": np.empty((128, 75), np.float32), "pool2": np.empty((128, 37), np.float32),
        }
        rc = lib().oracle_layer_taps(C.byref(self._w), _p(x), *[_p(out[k]) for k in (
            "conv1", "conv2", "pool1", "conv3", "conv4", "pool2")])
        assert rc == 0
        return out


def bf16_round(a):
    """fp32 array -> fp32 array of bf16-representable values (round-to-nearest-even; oracle_bf16_round)."""
    x = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    out = np.empty_like(x)
    lib().oracle_bf16_round_array(_p(x), C.c_int64(x.size), _p(out))
    return out


def bf16_bits(a):
    """bf16-representable fp32 array -> its uint16 bit patterns (the upper halves)."""
    x = np.ascontiguousarray(np.asarray(a), dtype=np.float32)
    return (x.view(np.uint32) >> 16).astype(np.uint16)


def bf16_from_bits(u):
    """uint16 bf16 bit patterns -> fp32 values."""
    return (np.ascontiguousarray(u, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def linear_rows(x, w, b, relu: bool):
    """y = act(x W^T + b) row by row, fp64 accumulate (src/contact_cnn.py:48-57)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    rc = lib().oracle_linear_rows(_p(x), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(w), _p(b), C.c_int(w.shape[0]),
                                  C.c_int(int(relu)), _p(y))
    assert rc == 0
    return y


def linear_rows_tree(x, w, b, relu: bool):
    """Bit-level model of libdce.so's fp32 Linear layers (oracle_linear_rows_tree): 4 K ranges, fmaf chains in the
    matrix pipe's K order, fixed combine -- what every fp32 FC kernel must return bit for bit on the same inputs."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    b = np.ascontiguousarray(b, dtype=np.float32)
    y = np.empty((x.shape[0], w.shape[0]), np.float32)
    rc = lib().oracle_linear_rows_tree(_p(x), C.c_int64(x.shape[0]), C.c_int(x.shape[1]), _p(w), _p(b), C.c_int(w.shape[0]),
                                       C.c_int(int(relu)), _p(y))
    assert rc == 0
    return y


def zscore_windows(seq):
    """(T,54) raw -> (T-149,150,54) z-scored windows (utils/data_handler.py:55-56)."""
    s = np.ascontiguousarray(seq, dtype=np.float32)
    T = s.shape[0]
    n = max(T - 149, 0)
    w = np.empty((n, 150, 54), np.float32)
    rc = lib().oracle_infer_sequence(None, _p(s), C.c_int64(T), _p(w), None, None, None)
    assert rc == 0
    return w


def argmax16(logits):
    lg = np.ascontiguousarray(logits, dtype=np.float32).reshape(-1, 16)
    return np.array([lib().oracle_argmax16(_p(lg[i:i + 1])) for i in range(lg.shape[0])], np.int32)


def decimal2binary(cls):
    """numpy restatement of reference src/inference_one_seq.py:59-62."""
    c = np.asarray(cls).astype(np.int64)
    return ((c[..., None] & np.array([8, 4, 2, 1])) != 0).astype(np.uint8)

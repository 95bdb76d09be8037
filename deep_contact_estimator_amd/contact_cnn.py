"""Host-side mirror of the reference model object (reference src/contact_cnn.py:7-66).

Same surface the reference scripts use --

    model = contact_cnn()
    model.load_state_dict(torch.load(path)['model_state_dict'])
    model = model.eval().to(device)
    output = model(input_data)            # (B,150,54) float32, z-scored  ->  (B,16) logits

-- but there are no torch.nn modules behind it: every call goes through the C ABI of
libdce.so (include/dce.h) into the hand-written gfx950 kernels.  Inputs may be numpy arrays
(host; staged by the library) or torch CUDA tensors (device pointers are passed straight
through and the launch is queued on torch's current stream).  No CPU fallback exists.
"""
from __future__ import annotations

import ctypes as C
import numbers
from typing import Mapping

import numpy as np

from . import _lib
from .synth import STATE_DICT_SHAPES

WINDOW, CHANNELS, CLASSES = 150, 54, 16
PRECISIONS = {"fp32": 0, "bf16_fc": 1}


def _is_torch(x) -> bool:
    return type(x).__module__.split(".")[0] == "torch"


def _device_index(device) -> int:
    """'cuda', 'cuda:1', torch.device, int -> HIP device ordinal.  'cpu' is refused."""
    if device is None:
        return 0
    if isinstance(device, numbers.Integral):          # int, numpy integer ids
        return int(device)
    s = str(device)
    if s.startswith("cpu"):
        raise RuntimeError("deep_contact_estimator_amd runs on MI355X only; there is no CPU path "
                           "(use the reference implementation for CPU inference)")
    if ":" in s:
        return int(s.split(":")[1])
    if _is_torch(device) or s in ("cuda", "hip"):
        try:
            import torch
            return torch.cuda.current_device()
        except Exception:
            return 0
    raise ValueError(f"unrecognised device {device!r}")


class contact_cnn:
    """contact_cnn on MI355X.  ``max_batch`` bounds the windows per kernel sequence (scratch
    is 29 KB per window: 0.95 GB at the default, of 288 GB); longer inputs are chunked inside the library."""

    def __init__(self, device=None, max_batch: int = 32768, precision: str = "fp32"):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self._device = device
        self._max_batch = int(max_batch)
        self._precision = precision
        self._state: dict[str, np.ndarray] = {}
        self._finalized = False

    # ---- lifecycle -------------------------------------------------------------------------
    def _ensure_ctx(self):
        if not self._ctx:
            dev = _device_index(self._device)
            ctx = C.c_void_p()
            _lib.check(self._lib.dce_create(C.byref(ctx), dev, self._max_batch), None)
            self._ctx = ctx
            self._dev_index = dev
        return self._ctx

    def close(self):
        if getattr(self, "_ctx", None):
            self._lib.dce_destroy(self._ctx)
            self._ctx = C.c_void_p()
        self._finalized = False             # a later call re-creates the ctx and uploads the weights again

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- nn.Module-like surface ------------------------------------------------------------
    def load_state_dict(self, state_dict: Mapping, strict: bool = True):
        """Accepts the reference's ``checkpoint['model_state_dict']`` (torch tensors) or a dict
        of numpy arrays with the same 14 keys (reference src/contact_cnn.py:8-58)."""
        expected = dict(STATE_DICT_SHAPES)
        missing = [k for k in expected if k not in state_dict]
        unexpected = [k for k in state_dict if k not in expected]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for contact_cnn: "
                               f"Missing key(s): {missing}. Unexpected key(s): {unexpected}.")
        for k, shape in expected.items():
            if k not in state_dict:
                continue
            v = state_dict[k]
            if _is_torch(v):
                v = v.detach().cpu().numpy()
            a = np.ascontiguousarray(np.asarray(v), dtype=np.float32)
            if a.shape != shape:
                raise RuntimeError(f"size mismatch for {k}: checkpoint {a.shape}, model {shape}")
            self._state[k] = a
        self._finalized = False
        return self

    def state_dict(self):
        return dict(self._state)

    def to(self, device):
        if self._ctx and _device_index(device) != self._dev_index:
            self.close()
            self._finalized = False
        self._device = device
        return self

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def eval(self):
        """Inference is the only mode (Dropout is identity); uploads/repacks the weights."""
        return self

    def train(self, mode: bool = True):
        if mode:
            raise RuntimeError("deep_contact_estimator_amd implements the inference path only")
        return self

    def _finalize(self):
        if self._finalized:
            return
        ctx = self._ensure_ctx()
        missing = [k for k, _ in STATE_DICT_SHAPES if k not in self._state]
        if missing:
            raise RuntimeError(f"load_state_dict first: missing {missing}")
        for k, _ in STATE_DICT_SHAPES:
            a = self._state[k]
            shape = (C.c_int64 * a.ndim)(*a.shape)
            _lib.check(self._lib.dce_load_weight(ctx, k.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim), ctx)
        _lib.check(self._lib.dce_finalize_weights(ctx, PRECISIONS[self._precision]), ctx)
        self._finalized = True

    # ---- compute ---------------------------------------------------------------------------
    def _run(self, x, raw_sequence: bool, want=("logits", "pred", "contacts")):
        self._finalize()
        ctx = self._ctx
        lib = self._lib
        if _is_torch(x) and x.is_cuda:
            import torch
            if x.device.index != self._dev_index:
                raise RuntimeError(f"input on cuda:{x.device.index}, model on cuda:{self._dev_index}")
            x = x.contiguous()
            if x.dtype != torch.float32:
                x = x.float()
            n = x.shape[0] - (WINDOW - 1) if raw_sequence else x.shape[0]
            n = max(n, 0)
            out = {}
            if "logits" in want: out["logits"] = torch.empty((n, CLASSES), dtype=torch.float32, device=x.device)
            if "pred" in want: out["pred"] = torch.empty((n,), dtype=torch.int32, device=x.device)
            if "contacts" in want: out["contacts"] = torch.empty((n, 4), dtype=torch.uint8, device=x.device)
            ptr = lambda k: C.c_void_p(out[k].data_ptr()) if k in out and n > 0 else None
            _lib.check(lib.dce_set_stream(ctx, C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream), 0), ctx)
            if raw_sequence:
                rc = lib.dce_infer_sequence(ctx, C.c_void_p(x.data_ptr()), x.shape[0], WINDOW, 1,
                                            ptr("logits"), ptr("pred"), ptr("contacts"))
            else:
                rc = lib.dce_forward_windows(ctx, C.c_void_p(x.data_ptr()), n, 1,
                                             ptr("logits"), ptr("pred"), ptr("contacts"))
            _lib.check(rc, ctx)
            return out
        was_torch = _is_torch(x)
        a = x.detach().numpy() if was_torch else np.asarray(x)
        a = np.ascontiguousarray(a, dtype=np.float32)
        n = a.shape[0] - (WINDOW - 1) if raw_sequence else a.shape[0]
        n = max(n, 0)
        out = {}
        if "logits" in want: out["logits"] = np.empty((n, CLASSES), np.float32)
        if "pred" in want: out["pred"] = np.empty((n,), np.int32)
        if "contacts" in want: out["contacts"] = np.empty((n, 4), np.uint8)
        ptr = lambda k: out[k].ctypes.data_as(C.c_void_p) if k in out and n > 0 else None
        _lib.check(lib.dce_set_stream(ctx, None, 1), ctx)
        if raw_sequence:
            rc = lib.dce_infer_sequence(ctx, a.ctypes.data_as(C.c_void_p), a.shape[0], WINDOW, 0,
                                        ptr("logits"), ptr("pred"), ptr("contacts"))
        else:
            rc = lib.dce_forward_windows(ctx, a.ctypes.data_as(C.c_void_p), n, 0,
                                         ptr("logits"), ptr("pred"), ptr("contacts"))
        _lib.check(rc, ctx)
        if was_torch:
            import torch
            out = {k: torch.from_numpy(v) for k, v in out.items()}
        return out

    @staticmethod
    def _check_windows(x):
        if x.ndim != 3 or tuple(x.shape[1:]) != (WINDOW, CHANNELS):
            raise RuntimeError(f"expected input (B,{WINDOW},{CHANNELS}), got {tuple(x.shape)}")

    def __call__(self, x):
        """model(input_data): (B,150,54) z-scored windows -> (B,16) logits (src/contact_cnn.py:60-66)."""
        self._check_windows(x)
        return self._run(x, False, want=("logits",))["logits"]

    forward = __call__

    def predict(self, x):
        """model(x) + torch.max(output,1) + decimal2binary in one pass
        (src/inference_one_seq.py:25-27) -> dict(logits, pred, contacts)."""
        self._check_windows(x)
        return self._run(x, False)

    def infer_sequence(self, seq):
        """contact_dataset + DataLoader + inference() fused: raw (T,54) sequence -> dict(logits
        (T-149,16), pred (T-149,), contacts (T-149,4)); row j belongs to data row j+149
        (utils/data_handler.py:24,55-57; src/inference_one_seq.py:19-30)."""
        if seq.ndim != 2 or seq.shape[1] != CHANNELS:
            raise RuntimeError(f"expected a (T,{CHANNELS}) sequence, got {tuple(seq.shape)}")
        return self._run(seq, True)

    def forward_taps(self, x):
        """Parity-test hook: numpy (n,150,54) -> dict(feat, h1, h2, logits) as numpy."""
        self._finalize()
        a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        self._check_windows(a)
        n = a.shape[0]
        out = {"h2": np.empty((n, 512), np.float32), "logits": np.empty((n, CLASSES), np.float32)}
        if self._precision == "fp32":          # feat / h1 are bf16 scratch in the bf16-FC mode
            out["feat"] = np.empty((n, 4736), np.float32)
            out["h1"] = np.empty((n, 2048), np.float32)
        p = lambda k: out[k].ctypes.data_as(C.c_void_p) if k in out else None
        _lib.check(self._lib.dce_set_stream(self._ctx, None, 1), self._ctx)
        _lib.check(self._lib.dce_forward_taps(self._ctx, a.ctypes.data_as(C.c_void_p), n, 0,
                                              p("feat"), p("h1"), p("h2"), p("logits")), self._ctx)
        return out

    def zscore_windows(self, seq, first: int = 0, n: int | None = None):
        """contact_dataset.__getitem__ for windows [first, first+n) (utils/data_handler.py:55-56)."""
        ctx = self._ensure_ctx()
        T = seq.shape[0]
        if n is None:
            n = T - (WINDOW - 1) - first
        if _is_torch(seq) and seq.is_cuda:
            import torch
            seq = seq.contiguous()
            out = torch.empty((n, WINDOW, CHANNELS), dtype=torch.float32, device=seq.device)
            _lib.check(self._lib.dce_set_stream(ctx, C.c_void_p(torch.cuda.current_stream(seq.device).cuda_stream), 0), ctx)
            _lib.check(self._lib.dce_zscore_windows(ctx, C.c_void_p(seq.data_ptr()), T, first, n, 1,
                                                    C.c_void_p(out.data_ptr()) if n > 0 else None), ctx)
            return out
        a = np.ascontiguousarray(seq.numpy() if _is_torch(seq) else np.asarray(seq), dtype=np.float32)
        out = np.empty((n, WINDOW, CHANNELS), np.float32)
        _lib.check(self._lib.dce_set_stream(ctx, None, 1), ctx)
        _lib.check(self._lib.dce_zscore_windows(ctx, a.ctypes.data_as(C.c_void_p), T, first, n, 0,
                                                out.ctypes.data_as(C.c_void_p)), ctx)
        return out

    def online_reset(self):
        """Empty the device-resident sample ring of the online mode."""
        _lib.check(self._lib.dce_online_reset(self._ensure_ctx()), self._ctx)

    def online_push(self, sample):
        """Online mode: append one (54,) sample; once 150 samples are in, returns
        (logits (16,), pred int, contacts (4,) u8) for the newest window, else None."""
        self._finalize()
        s = np.ascontiguousarray(np.asarray(sample), dtype=np.float32).reshape(-1)
        if s.shape[0] != CHANNELS:
            raise RuntimeError(f"expected a ({CHANNELS},) sample, got {s.shape}")
        logits = np.empty(CLASSES, np.float32); pred = np.empty(1, np.int32); contacts = np.empty(4, np.uint8)
        _lib.check(self._lib.dce_set_stream(self._ctx, None, 1), self._ctx)
        rc = self._lib.dce_online_push(self._ctx, s.ctypes.data_as(C.c_void_p), logits.ctypes.data_as(C.c_void_p),
                                       pred.ctypes.data_as(C.c_void_p), contacts.ctypes.data_as(C.c_void_p))
        if rc < 0:
            _lib.check(rc, self._ctx)
        return (logits, int(pred[0]), contacts) if rc == 1 else None

    def confusion_counts(self, pred, labels, counts=None):
        """Accumulate the 16x16 confusion counts C[gt][pred] on the device (dce_confusion_counts).
        pred: (n,) int32 numpy / CUDA tensor from predict() / infer_sequence(); labels: (n,) or
        (n,1) int64.  Returns (and, if given, updates) `counts`: (16,16) int64, same kind as pred."""
        ctx = self._ensure_ctx()
        if _is_torch(pred) and pred.is_cuda:
            import torch
            pred = pred.contiguous().to(torch.int32)
            labels = labels.to(pred.device).contiguous().to(torch.int64).reshape(-1)
            if counts is None:
                counts = torch.zeros((16, 16), dtype=torch.int64, device=pred.device)
            _lib.check(self._lib.dce_set_stream(ctx, C.c_void_p(torch.cuda.current_stream(pred.device).cuda_stream), 0), ctx)
            _lib.check(self._lib.dce_confusion_counts(ctx, C.c_void_p(pred.data_ptr()), C.c_void_p(labels.data_ptr()),
                                                      pred.shape[0], 1, C.c_void_p(counts.data_ptr())), ctx)
            return counts
        p = np.ascontiguousarray(pred.numpy() if _is_torch(pred) else pred, dtype=np.int32).reshape(-1)
        g = np.ascontiguousarray(labels.numpy() if _is_torch(labels) else labels, dtype=np.int64).reshape(-1)
        if counts is None:
            counts = np.zeros((16, 16), np.int64)
        _lib.check(self._lib.dce_set_stream(ctx, None, 1), ctx)
        _lib.check(self._lib.dce_confusion_counts(ctx, p.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p),
                                                  p.shape[0], 0, counts.ctypes.data_as(C.c_void_p)), ctx)
        return counts

    # ---- profiling (bench.py) ---------------------------------------------------------------
    def profile(self, every: int = 1):
        """Time the four kernels of every `every`-th kernel sequence with HIP events (0 = off)."""
        _lib.check(self._lib.dce_profile_enable(self._ensure_ctx(), int(every)), self._ctx)

    def profile_read(self, reset: bool = True):
        ms = (C.c_double * 4)()
        cnt = (C.c_int64 * 4)()
        _lib.check(self._lib.dce_profile_read(self._ctx, ms, cnt, int(reset)), self._ctx)
        names = ("conv_stack", "fc1_gemm", "fc2_gemm", "fc3_tail")
        return {k: {"ms": ms[i], "launches": cnt[i]} for i, k in enumerate(names)}

    def sync(self):
        if self._ctx:
            _lib.check(self._lib.dce_sync(self._ctx), self._ctx)


def load_checkpoint(path: str):
    """The reference's checkpoint schema (src/train.py:145-153): a torch-pickled dict with
    'model_state_dict'; also accepts a .npz of the 14 arrays."""
    if path.endswith(".npz"):
        z = np.load(path)
        return {k: z[k] for k in z.files}
    import torch
    ckpt = torch.load(path, map_location="cpu")
    return ckpt["model_state_dict"] if "model_state_dict" in ckpt else ckpt

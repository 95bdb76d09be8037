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
PACKED_ROW = 68                        # include/dce.h DCE_PACKED_ROW: 16 fp32 logits + 4 contact bits
PRECISIONS = {"fp32": 0, "bf16_fc": 1, "fp32_split": 2, "fp32_f16x2": 3}   # fp32_split: fc.0 on three-term bf16 operands (include/dce.h DCE_FP32_SPLIT)


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

    def __init__(self, device=None, max_batch: int = 32768, precision: str = "fp32", tune=None):
        """tune: the A/B switches of DESIGN.md's appendix for THIS model -- a dict {key: int} or "key=value,key=value"
        (include/dce.h dce_create_ex); None: the environment variable DCE_TUNE, else the defaults."""
        self._lib = _lib.load()
        self._tune = tune
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
            _lib.check(self._lib.dce_create_ex(C.byref(ctx), dev, self._max_batch, _lib.tune_spec(self._tune)), None)
            self._ctx = ctx
            self._dev_index = dev
        return self._ctx

    def close(self):
        self._online_stream_own = False
        if getattr(self, "_ctx", None) and getattr(self, "_ctx_abandoned", False):
            self._ctx = C.c_void_p()          # a thread is still inside dce_comm_init on it (distributed.comm_bootstrap timed out): not ours to destroy
        if getattr(self, "_ctx", None):
            self._lib.dce_destroy(self._ctx)
            self._ctx = C.c_void_p()
            self.comm_world = 0
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
            self._online_stream_own = False
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

    # ---- packed results + the multi-GPU exchange (include/dce.h: dce_*_packed, dce_comm_*, dce_gather_results) ----
    def _torch_stream(self, t):
        import torch
        _lib.check(self._lib.dce_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream), 0), self._ctx)
        self._online_stream_own = False

    def _run_packed(self, x, raw_sequence: bool, out=None):
        """-> (n,68) uint8 rows: 16 fp32 logits + 4 contact bits per window, written by the last kernel of the path.
        CUDA tensor in -> CUDA tensor out (`out` may name a pre-allocated one); numpy in -> numpy out."""
        self._finalize()
        ctx, lib = self._ctx, self._lib
        if _is_torch(x) and x.is_cuda:
            import torch
            x = x.contiguous()
            if x.dtype != torch.float32:
                x = x.float()
            n = max(x.shape[0] - (WINDOW - 1) if raw_sequence else x.shape[0], 0)
            if out is None:
                out = torch.empty((n, PACKED_ROW), dtype=torch.uint8, device=x.device)
            elif tuple(out.shape) != (n, PACKED_ROW) or out.dtype != torch.uint8 or not out.is_contiguous() or out.data_ptr() % 4:
                raise RuntimeError(f"out must be a contiguous, 4-byte aligned ({n},{PACKED_ROW}) uint8 tensor")
            self._torch_stream(x)
            dst = C.c_void_p(out.data_ptr()) if n > 0 else None
            if raw_sequence:
                rc = lib.dce_infer_sequence_packed(ctx, C.c_void_p(x.data_ptr()), x.shape[0], WINDOW, 1, dst)
            else:
                rc = lib.dce_forward_windows_packed(ctx, C.c_void_p(x.data_ptr()), n, 1, dst)
            _lib.check(rc, ctx)
            return out
        a = np.ascontiguousarray(x.detach().numpy() if _is_torch(x) else np.asarray(x), dtype=np.float32)
        n = max(a.shape[0] - (WINDOW - 1) if raw_sequence else a.shape[0], 0)
        out = np.empty((n, PACKED_ROW), np.uint8)
        _lib.check(lib.dce_set_stream(ctx, None, 1), ctx)
        dst = out.ctypes.data_as(C.c_void_p) if n > 0 else None
        if raw_sequence:
            rc = lib.dce_infer_sequence_packed(ctx, a.ctypes.data_as(C.c_void_p), a.shape[0], WINDOW, 0, dst)
        else:
            rc = lib.dce_forward_windows_packed(ctx, a.ctypes.data_as(C.c_void_p), n, 0, dst)
        _lib.check(rc, ctx)
        return out

    def predict_packed(self, x, out=None):
        """predict() with the results as (B,68)-byte rows (the gather's wire format)."""
        self._check_windows(x)
        return self._run_packed(x, False, out)

    def infer_sequence_packed(self, seq, out=None):
        """infer_sequence() with the results as (T-149,68)-byte rows."""
        if seq.ndim != 2 or seq.shape[1] != CHANNELS:
            raise RuntimeError(f"expected a (T,{CHANNELS}) sequence, got {tuple(seq.shape)}")
        return self._run_packed(seq, True, out)

    def unpack_results(self, packed):
        """(n,68) packed rows -> dict(logits (n,16) f32, pred (n,) i32, contacts (n,4) u8), same kind as `packed`."""
        ctx = self._ensure_ctx()
        n = packed.shape[0]
        if _is_torch(packed) and packed.is_cuda:
            import torch
            packed = packed.contiguous()
            out = {"logits": torch.empty((n, CLASSES), dtype=torch.float32, device=packed.device),
                   "pred": torch.empty((n,), dtype=torch.int32, device=packed.device),
                   "contacts": torch.empty((n, 4), dtype=torch.uint8, device=packed.device)}
            self._torch_stream(packed)
            if n:
                _lib.check(self._lib.dce_unpack_results(ctx, C.c_void_p(packed.data_ptr()), n, 1, C.c_void_p(out["logits"].data_ptr()),
                                                        C.c_void_p(out["pred"].data_ptr()), C.c_void_p(out["contacts"].data_ptr())), ctx)
            return out
        a = np.ascontiguousarray(packed.numpy() if _is_torch(packed) else packed, dtype=np.uint8)
        out = {"logits": np.empty((n, CLASSES), np.float32), "pred": np.empty((n,), np.int32), "contacts": np.empty((n, 4), np.uint8)}
        if n:
            _lib.check(self._lib.dce_unpack_results(ctx, a.ctypes.data_as(C.c_void_p), n, 0, out["logits"].ctypes.data_as(C.c_void_p),
                                                    out["pred"].ctypes.data_as(C.c_void_p), out["contacts"].ctypes.data_as(C.c_void_p)), ctx)
        return out

    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: a fresh ncclUniqueId (128 bytes) for comm_init; carry it to the other ranks by any means."""
        _lib.use_process_rccl()
        buf = (C.c_uint8 * 128)()
        _lib.check(_lib.load().dce_comm_get_unique_id(buf), None)
        return bytes(buf)

    def comm_init(self, rank: int, world: int, unique_id: bytes):
        """Join the RCCL communicator of the node's ranks (collective: blocks until all `world` ranks call)."""
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        _lib.use_process_rccl()
        ctx = self._ensure_ctx()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _lib.check(self._lib.dce_comm_init(ctx, int(rank), int(world), buf), ctx)
        self.comm_rank, self.comm_world = int(rank), int(world)
        return self

    comm_rank, comm_world = 0, 0          # comm_world > 0: this model holds an RCCL communicator

    def comm_info(self) -> dict:
        """What RCCL itself reports for the communicator, and the library that was bound."""
        r, w, v = C.c_int(), C.c_int(), C.c_int()
        name = C.create_string_buffer(512)
        _lib.check(self._lib.dce_comm_info(self._ctx, C.byref(r), C.byref(w), C.byref(v), name, 512), self._ctx)
        return {"rank": r.value, "world": w.value, "rccl_version": v.value, "library": name.value.decode()}

    def gather_results(self, packed, sizes=None, root: int = 0, out=None, async_: bool = False):
        """ONE RCCL gather of every rank's (n_r,68) packed CUDA rows to `root` (dce_gather_results), on the library's
        communication stream behind the kernels that produced them.  sizes[r] = rows of rank r (None: all equal).
        Returns the (sum,68) tensor on the root (stream-ordered; `out` may name it), None elsewhere.  async_: the
        gather overlaps later kernels; alternate two `packed`/`out` buffers and call comm_sync() before reading."""
        import torch
        if not self.comm_world:
            raise RuntimeError("gather_results: call comm_init first")
        packed = packed.contiguous()
        n = packed.shape[0]
        total = sum(sizes) if sizes is not None else n * self.comm_world
        if self.comm_rank == root:
            if out is None:
                out = torch.empty((total, PACKED_ROW), dtype=torch.uint8, device=packed.device)
            elif tuple(out.shape) != (total, PACKED_ROW) or not out.is_contiguous():
                raise RuntimeError(f"out must be a contiguous ({total},{PACKED_ROW}) uint8 tensor")
        rows = (C.c_int64 * self.comm_world)(*[int(v) for v in sizes]) if sizes is not None else None
        self._torch_stream(packed)
        _lib.check(self._lib.dce_gather_results(self._ctx, C.c_void_p(packed.data_ptr()) if n else None, n,
                                                C.c_void_p(out.data_ptr()) if (self.comm_rank == root and total) else None,
                                                rows, int(root), int(bool(async_))), self._ctx)
        return out if self.comm_rank == root else None

    def allreduce_counts(self, counts):
        """Sum the (16,16) int64 confusion counts over the ranks (ncclAllReduce), in place."""
        if _is_torch(counts) and counts.is_cuda:
            self._torch_stream(counts)
            _lib.check(self._lib.dce_allreduce_counts(self._ctx, C.c_void_p(counts.data_ptr()), 1), self._ctx)
            return counts
        a = counts.numpy() if _is_torch(counts) else counts
        _lib.check(self._lib.dce_allreduce_counts(self._ctx, a.ctypes.data_as(C.c_void_p), 0), self._ctx)
        return counts

    def comm_sync(self):
        _lib.check(self._lib.dce_comm_sync(self._ctx), self._ctx)

    def comm_destroy(self):
        if self.comm_world and self._ctx:
            _lib.check(self._lib.dce_comm_destroy(self._ctx), self._ctx)
        self.comm_world = 0

    def forward_taps(self, x):
        """Parity-test hook: numpy (n,150,54) -> dict(feat, h1, h2, logits) as numpy (feat / h1 are uint16 bf16 bit
        patterns in the bf16_fc precision)."""
        self._finalize()
        a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        self._check_windows(a)
        n = a.shape[0]
        out = {"h2": np.empty((n, 512), np.float32), "logits": np.empty((n, CLASSES), np.float32)}
        act = np.uint16 if self._precision == "bf16_fc" else np.float32     # bf16-FC mode: the bf16 bit patterns
        out["feat"] = np.empty((n, 4736), act)
        out["h1"] = np.empty((n, 2048), act)
        p = lambda k: out[k].ctypes.data_as(C.c_void_p) if k in out else None
        _lib.check(self._lib.dce_set_stream(self._ctx, None, 1), self._ctx)
        _lib.check(self._lib.dce_forward_taps(self._ctx, a.ctypes.data_as(C.c_void_p), n, 0,
                                              p("feat"), p("h1"), p("h2"), p("logits")), self._ctx)
        return out

    CONV_KERNELS = {"wino2": 0, "wino1x8": 1, "half": 2, "quarter": 3, "direct": 4, "wino1x4": 5, "wino2rt4": 6, "x3": 7, "h2": 8}

    def conv_layer_taps(self, x, kernel="wino2"):
        """Parity-test hook (dce_conv_layer_taps): numpy (n<=64,150,54) pre-normalised windows through ONE named conv
        kernel family -> dict(conv1 (n,64,150), conv2 (n,64,150), pool1 (n,64,75), conv3 (n,128,75), conv4 (n,128,75),
        feat (n,4736)), post-ReLU, PyTorch layout -- the reference's forward-hook taps (tests/golden/make_golden.py)."""
        self._finalize()
        a = np.ascontiguousarray(np.asarray(x), dtype=np.float32)
        self._check_windows(a)
        n = a.shape[0]
        out = {"conv1": np.empty((n, 64, 150), np.float32), "conv2": np.empty((n, 64, 150), np.float32),
               "pool1": np.empty((n, 64, 75), np.float32), "conv3": np.empty((n, 128, 75), np.float32),
               "conv4": np.empty((n, 128, 75), np.float32), "feat": np.empty((n, 4736), np.float32)}
        _lib.check(self._lib.dce_set_stream(self._ctx, None, 1), self._ctx)
        _lib.check(self._lib.dce_conv_layer_taps(self._ctx, a.ctypes.data_as(C.c_void_p), n, self.CONV_KERNELS.get(kernel, kernel),
                                                 *[out[k].ctypes.data_as(C.c_void_p) for k in ("conv1", "conv2", "pool1", "conv3", "conv4", "feat")]),
                   self._ctx)
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
            self._online_stream_own = False
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
        (logits (16,), pred int, contacts (4,) u8) for the newest window, else None.
        (The call is on the latency path of a robot's control loop: buffers and their ctypes pointers are made once.)"""
        self._finalize()
        ob = getattr(self, "_online_bufs", None)
        if ob is None:
            s = np.empty(CHANNELS, np.float32); lg = np.empty(CLASSES, np.float32); pr = np.empty(1, np.int32); ct = np.empty(4, np.uint8)
            ob = self._online_bufs = (s, lg, pr, ct, s.ctypes.data_as(C.c_void_p), lg.ctypes.data_as(C.c_void_p),
                                      pr.ctypes.data_as(C.c_void_p), ct.ctypes.data_as(C.c_void_p))
        s, lg, pr, ct, ps, pl, pp, pc = ob
        a = np.asarray(sample)
        if a.size != CHANNELS:
            raise RuntimeError(f"expected a ({CHANNELS},) sample, got {a.shape}")
        s[:] = a.reshape(-1)
        if not getattr(self, "_online_stream_own", False):
            _lib.check(self._lib.dce_set_stream(self._ctx, None, 1), self._ctx)
            self._online_stream_own = True
        rc = self._lib.dce_online_push(self._ctx, ps, pl, pp, pc)
        if rc < 0:
            _lib.check(rc, self._ctx)
        return (lg.copy(), int(pr[0]), ct.copy()) if rc == 1 else None

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
            self._online_stream_own = False
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

    def split_guard(self) -> dict:
        """fp32_split's range guard for this model (dce_split_guard_info): bounds from the checkpoint, whether it was refused,
        the per-window limits of predict(), and how often the guard fired."""
        self._finalize()
        g = _lib.SplitGuard()
        _lib.check(self._lib.dce_split_guard_info(self._ctx, C.byref(g)), self._ctx)
        return {"enabled": bool(g.enabled), "refused": bool(g.refused), "x_hi": float(g.x_hi), "x_lo": float(g.x_lo), "z_max": float(g.z_max),
                "gain": list(g.gain), "offs": list(g.offs), "guarded_launches": int(g.guarded_launches),
                "windows_out_of_range": int(g.windows_out_of_range), "fallbacks_run": int(g.fallbacks_run), "reason": g.reason.decode()}

    def last_plan(self) -> list[str]:
        """Kernel families launched by this model's most recent kernel sequence, in launch order (dce_last_plan)."""
        buf = C.create_string_buffer(1024)
        _lib.check(self._lib.dce_last_plan(self._ensure_ctx(), buf, 1024), self._ctx)
        return buf.value.decode().split()

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

"""Sharding the sliding windows of one sequence over the GPUs of a node.

Windows are independent (no BatchNorm, Dropout off in eval: reference utils/data_handler.py:55-57
uses only rows [i, i+150)), so rank g of G takes a contiguous range of windows and therefore the
sequence rows of that range plus a 149-row halo; weights are replicated.  There is no collective
inside the model.  The one exchange the path has is collecting the results: ONE RCCL gather of
(n_g,68)-byte rows -- the (n_g,16) fp32 logits and the (n_g,4) u8 contacts side by side, written in
that form by the last kernel of the path -- to rank 0, point-to-point over xGMI, each peer on its own
link.  Shard sizes are a pure function of (n, world): nothing but the payload is ever exchanged.

Transport.  On the GPUs the gather is issued by libdce.so itself (include/dce.h dce_gather_results:
ncclGather / one group of ncclSend+ncclRecv on a communication stream of the ctx; the count all-reduce
of the accuracy epilogue is dce_allreduce_counts) -- pass the `model` that holds the communicator
(`comm_bootstrap`).  ``torch.distributed`` is the launcher-side plumbing only: rank environment,
host barriers, and the store that carries the 128-byte ncclUniqueId.  Without a communicator the same
functions exchange through ``torch.distributed`` ("gloo" in the CPU tests of this logic and when ranks
share one GPU, which RCCL refuses).
"""
from __future__ import annotations

WINDOW = 150


def shard_range(n_windows: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous, balanced window range [lo, hi) of `rank` (first n%world ranks get one extra)."""
    if n_windows <= 0:
        return 0, 0
    base, extra = divmod(n_windows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_rows(T: int, rank: int, world: int) -> tuple[int, int, int, int]:
    """-> (row_lo, row_hi, win_lo, win_hi): rows [row_lo,row_hi) of the (T,54) sequence hold
    exactly the windows [win_lo,win_hi) of this rank (149-row halo included)."""
    lo, hi = shard_range(T - WINDOW + 1, rank, world)
    if hi <= lo:
        return 0, 0, lo, hi
    return lo, hi + WINDOW - 1, lo, hi


def shard_sizes(n_windows: int, world: int) -> list[int]:
    """Rows every rank contributes: a pure function of (n_windows, world), so no size exchange."""
    return [hi - lo for lo, hi in (shard_range(n_windows, r, world) for r in range(world))]


PACK_COLS = 68      # one result row on the wire: 16 fp32 logits (64 B) + 4 contact bits (4 B)


def pack_results(out):
    """{'logits' (n,16) f32, 'contacts' (n,4) u8[, 'pred']} -> ONE (n,68) uint8 block, so the
    exchange is a single collective.  'pred' is not sent: it is the 4 contact bits read as a
    number (decimal2binary is a bijection on 0..15; reference src/inference_one_seq.py:59-62)."""
    import torch
    lg = out["logits"].contiguous()
    n = lg.shape[0]
    buf = torch.empty((n, PACK_COLS), dtype=torch.uint8, device=lg.device)
    if n == 0:                                           # a rank without windows (fewer windows than ranks): an empty block
        return buf                                       # (a zero-row tensor from numpy has stride 0 and cannot be re-viewed)
    buf[:, :64] = lg.view(torch.uint8).reshape(n, 64)
    buf[:, 64:] = out["contacts"]
    return buf


def unpack_results(buf):
    """Inverse of pack_results (bit-exact) -> {'logits', 'pred' (int32), 'contacts'}."""
    import torch
    n = buf.shape[0]
    logits = buf[:, :64].contiguous().view(torch.float32).reshape(n, 16)
    contacts = buf[:, 64:].contiguous()
    c = contacts.to(torch.int32)
    pred = c[:, 0] * 8 + c[:, 1] * 4 + c[:, 2] * 2 + c[:, 3]
    return {"logits": logits, "pred": pred, "contacts": contacts}


def gather_rows(t, sizes=None, group=None, dst: int = 0):
    """Gather per-rank row blocks to `dst`, concatenated in rank order, with ONE collective.
    `sizes[r]` = rows of rank r; for sharded windows it is shard_sizes(n, world), known to every
    rank without communication.  (sizes=None: every rank passes the same number of rows.)
    Blocks are padded to the longest one (shards differ by at most one row)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = t.device
    if dist.get_backend(group) == "gloo":     # CPU transport (tests; no xGMI): exchange host copies
        t = t.cpu()
    if sizes is None:
        sizes = [t.shape[0]] * world
    if t.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {t.shape[0]} rows, shard_sizes says {sizes[rank]}")
    nmax = max(sizes)
    pad = t
    if t.shape[0] < nmax:
        pad = torch.cat([t, t.new_zeros((nmax - t.shape[0],) + tuple(t.shape[1:]))], 0)
    pad = pad.contiguous()
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], 0).to(home)


ID_FILE_NONCE = 16      # bytes of job nonce in front of the 128-byte ncclUniqueId (the same file format as tests/c/abi_ranks.c)


def id_file_rendezvous(path: str, rank: int, make_id, timeout: float = 120.0) -> bytes:
    """The ncclUniqueId of rank 0 through a file, safe against a file an earlier job left behind: the payload is
    <16-byte nonce><128-byte id>, the nonce comes from DCE_COMM_NONCE (the launcher draws one per job; tools/launch_ranks.sh);
    rank 0 removes a left-over, writes a temporary file and renames it; the others accept only a file with THEIR nonce -- a
    stale id would otherwise send ncclCommInitRank waiting for peers that will never come -- and give up after `timeout`.
    Without DCE_COMM_NONCE the nonce is DERIVED from what the ranks of one job share and two jobs on a node -- or two runs of the same job,
    one after the other on the default port -- do not: the launcher's rendezvous endpoint (MASTER_ADDR:MASTER_PORT, TORCHELASTIC_RUN_ID)
    AND the parent process id (the ranks of one launcher are siblings; a relaunch has another launcher process).  Ranks that are not
    siblings -- started by hand from different shells, or next to tests/c/abi_ranks.c, whose nonce is all zeros when the variable is unset --
    have to set DCE_COMM_NONCE to the same value everywhere."""
    import hashlib
    import os
    import time
    explicit = os.environ.get("DCE_COMM_NONCE", "")
    if explicit:
        nonce = explicit.encode()[:ID_FILE_NONCE].ljust(ID_FILE_NONCE, b"\0")
    else:
        shared = "|".join(os.environ.get(k, "") for k in ("MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")) + f"|ppid:{os.getppid()}"
        nonce = hashlib.sha256(shared.encode()).digest()[:ID_FILE_NONCE]
    if rank == 0:
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        uid = make_id()
        tmp = f"{path}.tmp.{os.getpid()}"
        with open(tmp, "wb") as f:
            f.write(nonce + uid)
        os.replace(tmp, path)
        return uid
    t0 = time.time()
    while time.time() - t0 < timeout:
        try:
            with open(path, "rb") as f:
                buf = f.read()
            if len(buf) == ID_FILE_NONCE + 128 and buf[:ID_FILE_NONCE] == nonce:
                return buf[ID_FILE_NONCE:]
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise RuntimeError(f"no ncclUniqueId of this job appeared at {path} within {timeout:.0f} s: rank 0 did not write one, or wrote it under another nonce -- "
                       "DCE_COMM_NONCE must be the SAME on every rank (unset, it is derived from MASTER_ADDR / MASTER_PORT / TORCHELASTIC_RUN_ID and the "
                       "parent process id, so ranks that are not children of one launcher have to set it)")


_bootstraps = 0


def comm_bootstrap(model, rank: int, world: int, key: str = "dce_comm_id"):
    """Give `model` (one per rank) the node's RCCL communicator: rank 0 draws an ncclUniqueId and the 128 bytes travel
    through the store of the initialised torch.distributed group (TCP, no GPU collective), or -- with DCE_COMM_ID_FILE
    set and no process group -- through that file (id_file_rendezvous; removed by rank 0 once every rank has joined).
    dce_comm_init is collective; without a process group it runs under a deadline (DCE_COMM_TIMEOUT, default 180 s), so that
    a peer that never arrives ends in an error, not in a hang."""
    import os
    import threading
    global _bootstraps
    _bootstraps += 1                       # every rank calls in the same order: the n-th communicator of a process has its own key / file
    key = f"{key}_{_bootstraps}"
    path = os.environ.get("DCE_COMM_ID_FILE")
    if path:
        path = f"{path}.{_bootstraps}"
    timeout = float(os.environ.get("DCE_COMM_TIMEOUT", "180"))
    store = None
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            from torch.distributed.distributed_c10d import _get_default_store
            store = _get_default_store()
    except Exception:
        store = None
    if store is not None:
        if rank == 0:
            store.set(key, type(model).comm_unique_id())
        uid = bytes(store.get(key))
        model.comm_init(rank, world, uid)
        return model
    if path:
        uid = id_file_rendezvous(path, rank, type(model).comm_unique_id, timeout=min(timeout, 120.0))
        done, err = threading.Event(), []

        def init():
            try:
                model.comm_init(rank, world, uid)
            except Exception as e:                                # noqa: BLE001
                err.append(e)
            done.set()

        th = threading.Thread(target=init, daemon=True, name="dce-comm-init")
        th.start()
        if not done.wait(timeout):
            # the daemon thread is still inside ncclCommInitRank ON THIS CONTEXT: the context must not be destroyed under it.  The model
            # gives its context up (close() then leaves it alone: a leak, in a process that is about to report a failure anyway).
            model._ctx_abandoned = True
            raise RuntimeError(f"dce_comm_init did not return within {timeout:.0f} s: {world - 1} peer(s) expected through {path}")
        if err:
            raise err[0]
        if rank == 0:
            try:
                os.unlink(path)            # every rank has joined: nobody reads it any more, the next job starts clean
            except FileNotFoundError:
                pass
        return model
    if world == 1:
        model.comm_init(rank, world, type(model).comm_unique_id())
        return model
    raise RuntimeError("comm_bootstrap needs an initialised torch.distributed group or DCE_COMM_ID_FILE")


def comm_bootstrap_checked(model, rank: int, world: int, device, key: str = "dce_comm_id", timeout: float | None = None):
    """comm_bootstrap plus a first exchange whose CONTENT the root checks (four rows per rank, every byte = rank + 1,
    through dce_gather_results), both under a watchdog, and a verdict every rank of the torch.distributed group agrees on.
    Returns None when the library's communicator works on every rank; otherwise the reason, with this model's communicator
    dropped, so that the caller can run its exchange over torch.distributed instead and SAY so.  (A communicator that
    fails or hangs on one rank would otherwise take the whole job with it; DCE_COMM_TIMEOUT, default 180 s.)"""
    import os
    import threading
    import torch
    import torch.distributed as dist
    if timeout is None:
        timeout = float(os.environ.get("DCE_COMM_TIMEOUT", "180"))
    verdict = {}
    given_up = threading.Event()

    def attempt():
        try:
            comm_bootstrap(model, rank, world, key=key)
            if given_up.is_set():
                raise RuntimeError("answered after the deadline")
            if os.environ.get("DCE_COMM_SELFTEST") == "fail":       # tests: the path a failed first exchange takes
                raise RuntimeError("DCE_COMM_SELFTEST=fail")
            rows = 4
            mine = torch.full((rows, PACK_COLS), (rank + 1) & 0xFF, dtype=torch.uint8, device=device)
            got = model.gather_results(mine, None, root=0)
            model.comm_sync()
            if rank == 0:
                want = (torch.arange(world, device=device).repeat_interleave(rows) + 1).to(torch.uint8)
                if not bool((got == want[:, None]).all()):
                    raise RuntimeError("the first gather delivered wrong bytes")
            verdict["ok"] = True
        except Exception as e:                                    # noqa: BLE001 -- any failure means: fall back
            verdict["why"] = f"{type(e).__name__}: {e}"
        if given_up.is_set():                                     # answered after the deadline: nobody uses this communicator
            try:
                model.comm_destroy()
            except Exception:                                     # noqa: BLE001
                pass
            model.comm_world = 0

    th = threading.Thread(target=attempt, daemon=True, name="dce-comm-bootstrap")
    th.start()
    th.join(timeout)
    ok = bool(verdict.get("ok"))
    why = verdict.get("why") or ("" if ok else f"no answer within {timeout:.0f} s")
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32,
                        device=device if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if int(flag.item()):
        return None
    given_up.set()
    if not th.is_alive():
        try:
            model.comm_destroy()
        except Exception:                                         # noqa: BLE001
            pass
    model.comm_world = 0
    return why or "another rank failed"


def infer_sequence_sharded(run, seq, group=None, dst: int = 0, n_windows: int | None = None, row_lo: int = 0, model=None):
    """Run `run(seq_rows) -> {'logits','pred','contacts'}` (e.g. contact_cnn.infer_sequence) on
    this rank's shard of the sequence and gather the results to rank `dst` in window order with a
    single collective (logits + contact bits packed per row).  Returns the full dict on `dst`,
    None elsewhere.

    seq: the whole (T,54) sequence (identical on every rank, or at least valid on its own row
    range), or -- with `n_windows` = windows of the WHOLE sequence and `row_lo` = global index of
    seq's first row -- just this rank's rows (halo included), as when every rank generated or loaded
    only its own slice."""
    if model is not None and getattr(model, "comm_world", 0):
        # the product path on GPUs: packed rows straight from the tail kernel, ONE RCCL gather issued by libdce.so
        world, rank = model.comm_world, model.comm_rank
        if n_windows is None:
            n_windows = max(seq.shape[0] - WINDOW + 1, 0)
        sizes = shard_sizes(n_windows, world)
        r0, r1, _, _ = shard_rows(n_windows + WINDOW - 1 if n_windows > 0 else 0, rank, world)
        packed = model.infer_sequence_packed(seq[r0 - row_lo:r1 - row_lo])
        got = model.gather_results(packed, sizes, root=dst)
        return model.unpack_results(got) if rank == dst else None
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if n_windows is None:
        n_windows = max(seq.shape[0] - WINDOW + 1, 0)
    sizes = shard_sizes(n_windows, world)
    r0, r1, _, _ = shard_rows(n_windows + WINDOW - 1 if n_windows > 0 else 0, rank, world)
    out = run(seq[r0 - row_lo:r1 - row_lo])
    got = gather_rows(pack_results(out), sizes, group, dst)
    return unpack_results(got) if rank == dst else None


def confusion_sharded(run, count, seq, labels, group=None, model=None):
    """The accuracy epilogue of the reference's test loop (src/test.py:19-70,102-104) over the GPUs
    of a node: every rank classifies its shard of windows with ``run(seq_rows) -> {'pred', ...}``,
    forms the 16x16 counts of (label, prediction) pairs with ``count(pred, labels) -> (16,16) int64
    tensor`` (e.g. contact_cnn.confusion_counts) and the matrices are summed with ONE all-reduce
    (2 KB).  Window j carries ``labels[j + 149]`` (utils/data_handler.py:57).  Every rank gets the
    full matrix; every metric the reference prints is a function of it (metrics.py)."""
    import torch
    import torch.distributed as dist
    rccl = model is not None and getattr(model, "comm_world", 0)
    world = model.comm_world if rccl else dist.get_world_size(group)
    rank = model.comm_rank if rccl else dist.get_rank(group)
    r0, r1, w0, w1 = shard_rows(seq.shape[0], rank, world)
    if w1 > w0:
        pred = run(seq[r0:r1])["pred"]
        C = count(pred, labels[w0 + WINDOW - 1:w1 + WINDOW - 1])
        C = torch.as_tensor(C).to(torch.int64).reshape(16, 16).clone()
    else:                                      # more ranks than windows: this rank contributes zeros
        C = torch.zeros((16, 16), dtype=torch.int64, device=getattr(seq, "device", None))
    if rccl:
        return model.allreduce_counts(C.contiguous())        # ncclAllReduce issued by libdce.so (dce_allreduce_counts)
    if dist.get_backend(group) == "gloo":
        C = C.cpu()
    dist.all_reduce(C, op=dist.ReduceOp.SUM, group=group)
    return C


def init_from_env():
    """One process per GPU under ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE
    in the environment): bind this process to its GPU and join the RCCL ("nccl") group.
    -> (rank, world, local_rank); (0, 1, 0) and no process group when launched plainly."""
    import os
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("DCE_FORCE_DIST"):      # DCE_FORCE_DIST: a one-rank group (RCCL smoke test on one GPU)
        import torch
        import torch.distributed as dist
        # DCE_DIST_BACKEND=gloo: functional test of the multi-process path on a box with fewer GPUs
        # than ranks (RCCL refuses two ranks on one device); ranks then share GPUs round-robin
        backend = os.environ.get("DCE_DIST_BACKEND", "nccl")
        if backend != "nccl":
            local = local % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local))
            else:
                dist.init_process_group(backend)
    return rank, world, local


class AsyncRowGather:
    """The per-step result exchange of bench.py / a serving loop: gather every rank's (rows, cols)
    block to rank `dst` asynchronously, `depth` gathers in flight, so step i's gather rides behind
    step i+1's kernels.  The submitted tensor is kept alive until its gather has completed; on
    `dst`, `latest()` returns the most recently completed list of per-rank blocks."""

    def __init__(self, rows: int, cols: int, dtype, device, group=None, dst: int = 0, depth: int = 2):
        import torch
        import torch.distributed as dist
        self.group, self.dst, self.depth = group, dst, depth
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._host = dist.get_backend(group) == "gloo"      # CPU transport (functional tests only)
        if self._host:
            device = "cpu"
        self._bufs = [[torch.empty((rows, cols), dtype=dtype, device=device) for _ in range(self.world)]
                      if self.rank == dst else None for _ in range(depth)]
        self._handles = [None] * depth
        self._keep = [None] * depth
        self._i = 0
        self._done = None

    def _wait(self, s):
        if self._handles[s] is not None:
            self._handles[s].wait()
            self._handles[s] = None
            self._keep[s] = None
            self._done = s

    def submit(self, t):
        import torch.distributed as dist
        s = self._i % self.depth
        self._wait(s)                         # slot free again (its buffers may be overwritten)
        if self._host:
            t = t.cpu()
        self._keep[s] = t
        self._handles[s] = dist.gather(t, self._bufs[s], dst=self.dst, group=self.group, async_op=True)
        self._i += 1

    def drain(self):
        for k in range(self.depth):           # oldest first, so `latest` ends on the newest
            self._wait((self._i + k) % self.depth)

    def latest(self):
        return None if self._done is None or self.rank != self.dst else self._bufs[self._done]


class PackedStepGather:
    """The per-step result exchange of bench.py / a serving loop on RCCL through the C ABI: every step the model writes
    its (rows,68) packed results into one of two device buffers and libdce.so gathers them to rank `dst` on its
    communication stream (dce_gather_results, async), so step i's gather rides behind step i+1's kernels; the library
    orders step i+2's writes behind gather i.  `latest()` on `dst` = the (world*rows,68) block of the newest finished step."""

    def __init__(self, model, rows: int, device, dst: int = 0):
        import torch
        self.model, self.dst, self.rows = model, dst, rows
        self.send = [torch.empty((rows, PACK_COLS), dtype=torch.uint8, device=device) for _ in range(2)]
        self.recv = ([torch.empty((rows * model.comm_world, PACK_COLS), dtype=torch.uint8, device=device) for _ in range(2)]
                     if model.comm_rank == dst else [None, None])
        self._i = 0
        self.bytes_per_step = rows * PACK_COLS * model.comm_world       # what lands on the root per step

    def step(self, windows):
        s = self._i & 1
        packed = self.model.predict_packed(windows, out=self.send[s])
        self.model.gather_results(packed, None, root=self.dst, out=self.recv[s], async_=True)
        self._i += 1
        return packed

    def drain(self):
        self.model.comm_sync()

    def latest(self):
        return None if self._i == 0 else self.recv[(self._i - 1) & 1]

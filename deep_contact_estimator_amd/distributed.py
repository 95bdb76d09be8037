"""Sharding the sliding windows of one sequence over the GPUs of a node.

Windows are independent (no BatchNorm, Dropout off in eval: reference utils/data_handler.py:55-57
uses only rows [i, i+150)), so rank g of G takes a contiguous range of windows and therefore the
sequence rows of that range plus a 149-row halo; weights are replicated.  There is no collective
inside the model.  The one exchange the path has is collecting the results: an RCCL gather of the
(n_g,16) fp32 logits (and the (n_g,4) u8 contacts) to rank 0 -- point-to-point over xGMI, each
peer on its own link.  One process per GPU; ``torch.distributed`` backend "nccl" (= RCCL) on the
GPUs, "gloo" in the CPU tests of this logic.
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


def gather_rows(t, group=None, dst: int = 0):
    """Gather per-rank row blocks of unequal length to `dst`, concatenated in rank order.
    One size exchange (all_gather of a scalar) + one gather of the padded blocks."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = t.device
    if dist.get_backend(group) == "gloo":     # CPU transport (tests; no xGMI): exchange host copies
        t = t.cpu()
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
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


def infer_sequence_sharded(run, seq, group=None, dst: int = 0):
    """Run `run(seq_rows) -> {'logits','pred','contacts'}` (e.g. contact_cnn.infer_sequence) on
    this rank's shard of `seq` ((T,54), identical on every rank or at least valid on its own row
    range) and gather the results to rank `dst` in window order.  Returns the full dict on `dst`,
    None elsewhere."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    r0, r1, _, _ = shard_rows(seq.shape[0], rank, world)
    out = run(seq[r0:r1])
    res = {k: gather_rows(v, group, dst) for k, v in out.items()}
    return res if rank == dst else None


def confusion_sharded(run, count, seq, labels, group=None):
    """The accuracy epilogue of the reference's test loop (src/test.py:19-70,102-104) over the GPUs
    of a node: every rank classifies its shard of windows with ``run(seq_rows) -> {'pred', ...}``,
    forms the 16x16 counts of (label, prediction) pairs with ``count(pred, labels) -> (16,16) int64
    tensor`` (e.g. contact_cnn.confusion_counts) and the matrices are summed with ONE all-reduce
    (2 KB).  Window j carries ``labels[j + 149]`` (utils/data_handler.py:57).  Every rank gets the
    full matrix; every metric the reference prints is a function of it (metrics.py)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    r0, r1, w0, w1 = shard_rows(seq.shape[0], rank, world)
    if w1 > w0:
        pred = run(seq[r0:r1])["pred"]
        C = count(pred, labels[w0 + WINDOW - 1:w1 + WINDOW - 1])
        C = torch.as_tensor(C).to(torch.int64).reshape(16, 16).clone()
    else:                                      # more ranks than windows: this rank contributes zeros
        C = torch.zeros((16, 16), dtype=torch.int64, device=getattr(seq, "device", None))
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
    if world > 1:
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

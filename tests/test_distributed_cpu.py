"""CPU, world_size 2, gloo: the multi-GPU sharding + gather logic (one process per rank).
The per-rank compute is stood in for by the CPU oracle (tests may use it); what is under test is
that halo-sharded ranges + the gather reproduce the single-process result row for row."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_ranges_cover_exactly():
    from deep_contact_estimator_amd.distributed import shard_range, shard_rows
    for n in (0, 1, 7, 8, 9, 1000, 1_000_000, 8_000_000):
        for world in (1, 2, 3, 8):
            edges = [shard_range(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == max(n, 0)
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            sizes = [hi - lo for lo, hi in edges]
            assert max(sizes) - min(sizes) <= 1
    r0, r1, w0, w1 = shard_rows(1000 + 149, 1, 2)
    assert (w0, w1) == (500, 1000) and (r0, r1) == (500, 1149)       # 149-row halo
    assert shard_rows(100, 0, 2)[:2] == (0, 0)                       # shorter than one window


def test_pack_unpack_is_bit_exact():
    """The single-collective wire format: 16 fp32 logits + 4 contact bits per row, pred re-derived."""
    import torch
    from deep_contact_estimator_amd.distributed import pack_results, unpack_results, shard_sizes, PACK_COLS
    rng = np.random.default_rng(5)
    lg = rng.standard_normal((37, 16)).astype(np.float32)
    lg[3, 2] = np.nan; lg[4, 0] = np.inf; lg[5, 1] = -0.0
    pred = rng.integers(0, 16, 37).astype(np.int32)
    contacts = ((pred[:, None] & np.array([8, 4, 2, 1])) != 0).astype(np.uint8)
    buf = pack_results({"logits": torch.from_numpy(lg), "contacts": torch.from_numpy(contacts)})
    assert buf.shape == (37, PACK_COLS) and buf.dtype == torch.uint8
    back = unpack_results(buf)
    assert np.array_equal(back["logits"].numpy().view(np.uint32), lg.view(np.uint32))      # bit pattern, NaN included
    assert np.array_equal(back["contacts"].numpy(), contacts) and np.array_equal(back["pred"].numpy(), pred)
    empty = unpack_results(pack_results({"logits": torch.zeros((0, 16)), "contacts": torch.zeros((0, 4), dtype=torch.uint8)}))
    assert empty["logits"].shape == (0, 16) and empty["pred"].shape == (0,)
    assert shard_sizes(10, 4) == [3, 3, 2, 2] and shard_sizes(0, 3) == [0, 0, 0] and sum(shard_sizes(8_000_000, 8)) == 8_000_000


def test_shards_pack_and_unpack_for_every_world_size_up_to_eight():
    """SURVEY 8(e)'s partition for every world size 1..8 and window counts around the multiples of 8 (and configs[3]'s 8e6, and a
    count that leaves a remainder on an 8-GPU node): contiguous cover, sizes within one of each other, halo rows exactly 149 past
    the last window, empty shards when there are fewer windows than ranks; and the gather's wire format through those very sizes --
    per-rank blocks packed, concatenated in rank order, unpacked -- reproduces the single-process arrays bit for bit."""
    import torch
    from deep_contact_estimator_amd.distributed import shard_range, shard_rows, shard_sizes, pack_results, unpack_results
    rng = np.random.default_rng(8)
    counts = sorted({0, 1, 2, 7, 8, 9, 15, 16, 17, 23, 24, 25, 63, 64, 65, 1000, 1_000_003, 8_000_000} | {8 * k + d for k in (1, 5, 13) for d in (-1, 0, 1)})
    for world in range(1, 9):
        for n in counts:
            sizes = shard_sizes(n, world)
            assert len(sizes) == world and sum(sizes) == n and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
            lo = 0
            for r in range(world):
                a, b = shard_range(n, r, world)
                assert (a, b) == (lo, lo + sizes[r]) if n else (a, b) == (0, 0)
                r0, r1, w0, w1 = shard_rows(n + 149, r, world)
                assert (w0, w1) == (a, b) and ((r0, r1) == (a, b + 149) if b > a else (r0, r1) == (0, 0))
                lo += sizes[r]
        n = 8 * 13 + world - 3                                  # the wire format through these sizes
        lg = rng.standard_normal((n, 16)).astype(np.float32)
        pred = lg.argmax(1).astype(np.int32)
        contacts = ((pred[:, None] & np.array([8, 4, 2, 1])) != 0).astype(np.uint8)
        blocks, lo = [], 0
        for sz in shard_sizes(n, world):
            blocks.append(pack_results({"logits": torch.from_numpy(lg[lo:lo + sz]), "contacts": torch.from_numpy(contacts[lo:lo + sz])}))
            lo += sz
        back = unpack_results(torch.cat(blocks))
        assert np.array_equal(back["logits"].numpy(), lg) and np.array_equal(back["pred"].numpy(), pred) and np.array_equal(back["contacts"].numpy(), contacts)


def _worker(rank, world, port, T, out_path, local_slices=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from deep_contact_estimator_amd import synth
    from deep_contact_estimator_amd.distributed import infer_sequence_sharded, shard_rows
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    o = orc.Oracle(synth.make_state_dict(1, "uniform"))
    seq = torch.from_numpy(synth.make_sequence(T, 17).astype(np.float32))

    def run(rows):
        r = o.infer_sequence(rows.numpy())
        return {k: torch.from_numpy(v) for k, v in r.items()}

    if local_slices:      # every rank holds only its own rows (halo included), as in configs[3]
        r0, r1, _, _ = shard_rows(T, rank, world)
        res = infer_sequence_sharded(run, seq[r0:r1].clone(), dst=0, n_windows=T - 149, row_lo=r0)
    else:
        res = infer_sequence_sharded(run, seq, dst=0)
    if rank == 0:
        np.savez(out_path, **{k: v.numpy() for k, v in res.items()})
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
@pytest.mark.parametrize("T,local_slices", [(150 + 60, False), (150 + 2, False), (151, False),   # even split, tiny, one rank empty
                                            (150 + 61, True), (151, True)])                         # ranks hold only their rows
def test_two_rank_gather_matches_single_process(T, local_slices, world, tmp_path):
    import torch.multiprocessing as mp
    from deep_contact_estimator_amd import synth
    from oracle import oracle as orc
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npz")
    if world == 8 and (T, local_slices) not in ((150 + 60, False), (150 + 61, True), (150 + 2, False)):
        pytest.skip("the 8-rank rehearsal runs the ragged, the rank-local and the mostly-empty case")
    mp.spawn(_worker, args=(world, port, T, out, local_slices), nprocs=world, join=True)
    got = np.load(out)
    ref = orc.Oracle(synth.make_state_dict(1, "uniform")).infer_sequence(
        synth.make_sequence(T, 17).astype(np.float32))
    for k in ("logits", "pred", "contacts"):
        assert np.array_equal(got[k], ref[k]), k


def _gather_worker(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from deep_contact_estimator_amd.distributed import AsyncRowGather
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = AsyncRowGather(8, 16, torch.float32, "cpu", dst=0, depth=2)
    for step in range(5):                                   # more steps than slots: slots get reused
        g.submit(torch.full((8, 16), float(100 * step + rank)))
    g.drain()
    if rank == 0:
        got = torch.stack(g.latest()).numpy()
        np.save(out_path, got)
    else:
        assert g.latest() is None
    dist.barrier()
    dist.destroy_process_group()


def test_async_row_gather_two_ranks(tmp_path):
    """bench.py's per-step logits exchange (2 gathers in flight) delivers, on rank 0, every rank's
    block of the LAST submitted step."""
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "g.npy")
    mp.spawn(_gather_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert got.shape == (2, 8, 16)
    assert (got[0] == 400.0).all() and (got[1] == 401.0).all()


def _confusion_worker(rank, world, port, T, out_path):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from deep_contact_estimator_amd import synth, metrics
    from deep_contact_estimator_amd.distributed import confusion_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seq = torch.from_numpy(synth.make_sequence(T, 23).astype(np.float32))
    labels = torch.from_numpy(synth.make_labels(T, 5).reshape(-1))

    def run(rows):               # a stand-in classifier that depends on the window's own rows only
        x = rows.numpy()
        n = x.shape[0] - 149
        pred = np.array([int(abs(x[j:j + 150].sum()) * 7) % 16 for j in range(n)], np.int32)
        return {"pred": torch.from_numpy(pred)}

    def count(pred, lab):
        return torch.from_numpy(metrics.confusion16(pred.numpy(), lab.numpy()))

    C = confusion_sharded(run, count, seq, labels)
    np.save(out_path + f".{rank}.npy", C.numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("T", [150 + 75, 150])               # 76 windows over 2 ranks; 1 window (rank 1 empty)
def test_two_rank_confusion_matches_single_process(T, tmp_path):
    import torch.multiprocessing as mp
    from deep_contact_estimator_amd import synth, metrics
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "conf")
    mp.spawn(_confusion_worker, args=(2, port, T, out), nprocs=2, join=True)
    seq = synth.make_sequence(T, 23).astype(np.float32)
    labels = synth.make_labels(T, 5).reshape(-1)
    n = T - 149
    pred = np.array([int(abs(seq[j:j + 150].sum()) * 7) % 16 for j in range(n)], np.int32)
    want = metrics.confusion16(pred, labels[149:])
    for r in range(2):                                       # all-reduce: every rank holds the full matrix
        got = np.load(out + f".{r}.npy")
        assert got.sum() == n and np.array_equal(got, want)

"""CPU: round-3 host logic that needs no GPU -- the packed-row wire format, the build's source hash, the
multi-GPU entry points' argument checking."""
import ctypes as C
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from deep_contact_estimator_amd import build, _lib
    build.build()
    return _lib.load()


def test_host_unpack_is_the_inverse_of_the_wire_format(lib):
    """dce_unpack_results on HOST rows needs no device (ctx may be NULL): 16 little-endian fp32 logits + 4 contact
    bits per 68-byte row -- the same format distributed.pack_results builds for the gloo transport of the CPU tests."""
    import torch
    from deep_contact_estimator_amd.distributed import pack_results, PACK_COLS
    rng = np.random.default_rng(5)
    n = 1000
    lg = rng.standard_normal((n, 16)).astype(np.float32)
    lg[3, 2] = np.nan; lg[4, 0] = np.inf; lg[5, 1] = -0.0
    pred = rng.integers(0, 16, n).astype(np.int32)
    contacts = ((pred[:, None] & np.array([8, 4, 2, 1])) != 0).astype(np.uint8)
    buf = pack_results({"logits": torch.from_numpy(lg), "contacts": torch.from_numpy(contacts)}).numpy()
    assert buf.shape == (n, PACK_COLS) == (n, 68)
    out_l = np.empty((n, 16), np.float32); out_p = np.empty(n, np.int32); out_c = np.empty((n, 4), np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.dce_unpack_results(None, p(buf), n, 0, p(out_l), p(out_p), p(out_c)) == 0
    assert np.array_equal(out_l.view(np.uint32), lg.view(np.uint32))
    assert np.array_equal(out_p, pred) and np.array_equal(out_c, contacts)
    assert lib.dce_unpack_results(None, p(buf), n, 1, p(out_l), None, None) < 0      # device rows need a ctx
    assert lib.dce_unpack_results(None, None, 0, 0, None, None, None) == 0          # empty is fine


def test_comm_entry_points_reject_bad_calls_without_crashing(lib):
    assert lib.dce_comm_init(None, 0, 1, None) < 0
    assert lib.dce_gather_results(None, None, 0, None, None, 0, 0) < 0
    assert lib.dce_allreduce_counts(None, None, 0) < 0
    assert lib.dce_comm_sync(None) < 0 and lib.dce_comm_destroy(None) < 0
    assert lib.dce_comm_get_unique_id(None) < 0
    buf = C.create_string_buffer(64)
    assert lib.dce_last_plan(None, buf, 64) < 0


def test_build_records_the_hash_of_its_sources():
    """libdce.so.srchash ties measurements to a build: bench.py compares it with the hash stored in
    profiles/pmc_latest.json (roofline.traffic_stale)."""
    from deep_contact_estimator_amd import build
    build.build()
    assert build.built_hash() == build.source_hash() and len(build.source_hash()) == 64


def test_shard_sizes_drive_the_ragged_gather():
    """dce_gather_results takes rows_per_rank = shard_sizes(n, world): contiguous, balanced, differing by at most one."""
    from deep_contact_estimator_amd.distributed import shard_sizes, shard_range
    for n, w in ((8_000_000, 8), (1_000_003, 8), (5, 8), (0, 4)):
        s = shard_sizes(n, w)
        assert sum(s) == n and max(s) - min(s) <= 1 and s == sorted(s, reverse=True)
        assert [shard_range(n, r, w)[1] - shard_range(n, r, w)[0] for r in range(w)] == s

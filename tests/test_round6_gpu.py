"""GPU (-m gpu), round 6: the latency mode's MICRO-BATCH kernel (csrc/latency_mb.hip: 2 .. 32 windows in one kernel -- the reference's shipped
batch_size 30, config/test_params.yaml:9, src/test.py:83-104,126) against the ORACLE, and the chunk rule of the latency plans."""
import numpy as np
import pytest

from conftest import tol_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def sd():
    from deep_contact_estimator_amd import synth
    return synth.make_state_dict(1, "uniform")


def _model(sd, max_batch=64, tune=None):
    from deep_contact_estimator_amd import contact_cnn
    m = contact_cnn(device=0, max_batch=max_batch, tune=tune)
    m.load_state_dict(sd).eval()
    return m


def _argmax_contract(pred, ref):
    srt = np.sort(ref["logits"], axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3 * np.abs(ref["logits"]).max()
    assert np.array_equal(pred[safe], ref["pred"][safe])


@pytest.mark.parametrize("n", [2, 3, 8, 12, 13, 16, 17, 30, 31, 32])
def test_latency_micro_batch_vs_oracle(n, sd, orc):
    """option latency=1, a call of 2 .. 32 windows: ONE kernel of 256 co-resident workgroups (conv segments -> fc.0 tiles with register-resident
    weights on fp32 MFMAs -> fc.3 tiles + partial logits -> the ordered sum, argmax, contact bits), every layer handed over through flags.  Not
    the batch path's bits (a wave's K range is one fp32 chain, eight of them are added in order), but inside the fp32 contract of the ORACLE:
    pre-normalised windows and raw rows (z-score fused), host and device pointers, packed rows, a non-finite window in the middle of a batch, the
    same bytes on every repeat.  Sizes on both sides of every switch: one / two windows per MFMA row tile (16 | 17), two conv workgroups per
    segment or one (16 | 17), fc.3 on idle CUs or on the conv workgroups (12 | 13), the largest batch (32)."""
    import torch
    from deep_contact_estimator_amd import synth
    m = _model(sd, tune={"latency": 1})
    o = orc.Oracle(sd)
    seq = synth.make_sequence(150 + n - 1, seed=40 + n, kind="ar1").astype(np.float32)
    zw = orc.zscore_windows(seq)
    ref = o.forward_windows(zw)
    got = m.predict(zw)
    assert m.last_plan() == ["latency_mb"], m.last_plan()
    tol_ok(got["logits"], ref["logits"], f"latency mode, {n} pre-normalised windows in one kernel")
    _argmax_contract(got["pred"], ref)
    assert np.array_equal(got["contacts"], orc.decimal2binary(got["pred"]))
    # raw rows through dce_infer_sequence (T = 149 + n)
    gz = m.infer_sequence(seq)
    assert m.last_plan() == ["latency_mb_zs"], m.last_plan()
    tol_ok(gz["logits"], ref["logits"], f"latency mode, {n} raw windows in one kernel")
    _argmax_contract(gz["pred"], ref)
    # device pointers on torch's stream; packed rows; repeats are bit-stable
    xt = torch.from_numpy(zw).cuda()
    d1, d2 = m.predict(xt), m.predict(xt)
    torch.cuda.synchronize()
    assert np.array_equal(d1["logits"].cpu().numpy(), got["logits"]) and np.array_equal(d2["logits"].cpu().numpy(), got["logits"])
    assert np.array_equal(d1["pred"].cpu().numpy(), got["pred"]) and np.array_equal(d1["contacts"].cpu().numpy(), got["contacts"])
    pk = m.predict_packed(xt).cpu().numpy()
    assert np.array_equal(pk[:, :64].copy().view(np.float32), got["logits"]) and np.array_equal(pk[:, 64:], got["contacts"])
    # a non-finite window poisons its own row only (NaN logits, class 0: torch.max's index of the first NaN)
    bad = zw.copy(); k = n // 2; bad[k, 17, 3] = np.nan
    r = m.predict(bad)
    assert np.isnan(r["logits"][k]).all() and r["pred"][k] == 0
    keep = np.arange(n) != k
    assert np.array_equal(r["logits"][keep], got["logits"][keep])
    # and the next call is clean again
    assert np.array_equal(m.predict(zw)["logits"], got["logits"])
    m.close()


def test_latency_micro_batch_every_size_interleaved(sd, orc):
    """Every size 2 .. 32 on ONE context, in an order that alternates between the role layouts (two conv workgroups per segment | one, fc.3 on idle CUs | on
    the conv workgroups | on the fc.0 role, one | two row tiles with the second one behind a conv segment), pre-normalised windows and raw rows: each call
    against the ORACLE, and equal -- bit for bit -- to the same windows' rows when they were part of another call's size (a window's logits depend on its
    row tile's position only through the fixed order of the sums: rows are independent).  Flags of differently sized calls share one array: a stale flag
    of an earlier, larger call must never satisfy a later wait (the request's number does that)."""
    from deep_contact_estimator_amd import synth
    m = _model(sd, tune={"latency": 1})
    o = orc.Oracle(sd)
    seq = synth.make_sequence(150 + 32 - 1, seed=77, kind="ar1").astype(np.float32)
    zw = orc.zscore_windows(seq)
    ref = o.forward_windows(zw)
    sizes = [32, 2, 17, 16, 31, 3, 13, 30, 12, 24, 5, 18, 9, 29, 4, 20, 8, 27, 6, 22, 10, 25, 7, 19, 11, 28, 14, 21, 15, 26, 23]
    assert sorted(sizes) == list(range(2, 33))
    full = None
    for n in sizes:
        got = m.predict(zw[:n])
        assert m.last_plan() == ["latency_mb"], (n, m.last_plan())
        tol_ok(got["logits"], ref["logits"][:n], f"latency mode, {n} windows")
        _argmax_contract(got["pred"], {"logits": ref["logits"][:n], "pred": ref["pred"][:n]})
        assert np.array_equal(got["contacts"], orc.decimal2binary(got["pred"]))
        gz = m.infer_sequence(seq[:149 + n])
        assert m.last_plan() == ["latency_mb_zs"], (n, m.last_plan())
        tol_ok(gz["logits"], ref["logits"][:n], f"latency mode, {n} raw windows")
        if full is None:
            full = got["logits"].copy()
        else:                                                      # rows 0 .. 15 sit in the first row tile whatever n is: the same chains, the same bits
            k = min(n, 16)
            assert np.array_equal(got["logits"][:k], full[:k]), n
    m.close()


def test_latency_micro_batch_limits_and_switch(sd, orc):
    """33 windows and more take the batch path's kernels (bit-identical to a context without the option); latency_mb=0 keeps 2 .. 32 there too;
    one window stays on latency.hip's kernel; the micro-batch kernel and the one-window kernel alternate on one context without disturbing
    each other's request counters."""
    from deep_contact_estimator_amd import synth
    m, off, b = _model(sd, tune={"latency": 1}), _model(sd, tune={"latency": 1, "latency_mb": 0}), _model(sd)
    x = np.random.default_rng(5).standard_normal((40, 150, 54), dtype=np.float32)
    ref = orc.Oracle(sd).forward_windows(x)
    r33 = m.predict(x[:33])
    assert m.last_plan()[0] == "conv_wino_quarter" and np.array_equal(r33["logits"], b.predict(x[:33])["logits"])
    r8 = off.predict(x[:8])
    assert "latency_mb" not in off.last_plan() and np.array_equal(r8["logits"], b.predict(x[:8])["logits"])
    for rep in range(20):                                       # 1, 30, 1, 2, .. on one context
        for n in (1, 30, 1, 2, 17):
            lo = (rep * 3) % (40 - n)
            g = m.predict(x[lo:lo + n])
            assert m.last_plan() == (["latency_one"] if n == 1 else ["latency_mb"]), m.last_plan()
            tol_ok(g["logits"], ref["logits"][lo:lo + n], f"rep {rep}, {n} windows from {lo}")
    m.close(); off.close(); b.close()


def test_latency_micro_batch_hand_overs_are_never_stale(sd, orc):
    """The hand-overs of the one kernel (features, h1, partial logits: write-through stores, agent-scope loads, one flag per producer) under a
    changing load: 300 back-to-back calls whose size AND data change every call -- a consumer that saw a flag early or read a stale line of an
    earlier request would return that request's numbers -- every call against the ORACLE, while a second stream keeps the memory system busy
    half of the time."""
    import torch
    rng = np.random.default_rng(9)
    pool = rng.standard_normal((96, 150, 54), dtype=np.float32) * rng.uniform(0.5, 3.0, (96, 1, 1)).astype(np.float32)
    ref = orc.Oracle(sd).forward_windows(pool)
    m = _model(sd, tune={"latency": 1})
    xt = torch.from_numpy(pool).cuda()
    side = torch.cuda.Stream()
    junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
    outs = []
    for it in range(300):
        n = int(rng.integers(2, 33)); lo = int(rng.integers(0, 96 - n))
        if it % 2:
            with torch.cuda.stream(side):
                junk.add_(1.0)                                   # 256 MB read + written beside the kernel
        outs.append((lo, n, m.predict(xt[lo:lo + n])))
    torch.cuda.synchronize()
    for lo, n, o in outs:
        tol_ok(o["logits"].cpu().numpy(), ref["logits"][lo:lo + n], f"{n} windows from {lo}")
    m.close()


def test_latency_plans_serve_whole_calls_only(sd, orc):
    """include/dce.h: DCE_FP32 gives a window the same bits whatever the size of the call.  A latency=1 context keeps that INSIDE a call: a call
    that is cut into chunks (more windows than max_batch) runs the batch kernels for every chunk, also for a last chunk of one or a few windows
    (round 5's advice: 2 x max_batch + 1 windows returned its last window on the one-window kernel, with other bits)."""
    m, b = _model(sd, max_batch=16, tune={"latency": 1}), _model(sd, max_batch=16)
    x = np.random.default_rng(6).standard_normal((2 * 16 + 5, 150, 54), dtype=np.float32)
    for n in (33, 34, 37):
        g, r = m.predict(x[:n]), b.predict(x[:n])
        assert not any(k.startswith("latency") for k in m.last_plan()), m.last_plan()
        assert np.array_equal(g["logits"], r["logits"]) and np.array_equal(g["pred"], r["pred"])
    assert m.predict(x[:5])["logits"].shape == (5, 16) and m.last_plan() == ["latency_mb"]      # a whole call of five windows: the one kernel
    m.close(); b.close()


def test_fp32_split_is_an_alias_of_fp32_f16x2_in_the_product_library(sd, orc):
    """Round 6: DCE_FP32_SPLIT (three bf16 terms per operand, range-guarded) left the product library -- DCE_FP32_F16X2 holds the same contract at 1.13 - 2.0 x
    its speed at every launch size (profiles/r6h_retire_split_sweep.txt).  A caller that asks for it gets fp32_f16x2's kernels and bits, dce_last_plan and
    dce_split_guard_info say so; the three-term kernels and their tests run in the experiments build (tests/test_experiments_gpu.py)."""
    from conftest import has_experiments
    from deep_contact_estimator_amd import contact_cnn
    if has_experiments():
        pytest.skip("the experiments build runs the real three-term kernels")
    a = contact_cnn(device=0, max_batch=4096, precision="fp32_split"); a.load_state_dict(sd).eval()
    b = contact_cnn(device=0, max_batch=4096, precision="fp32_f16x2"); b.load_state_dict(sd).eval()
    for n in (64, 700, 4096):
        x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
        ra, rb = a.predict(x), b.predict(x)
        assert a.last_plan()[0] == "fp32_split_is_fp32_f16x2" and a.last_plan()[1:] == b.last_plan(), (a.last_plan(), b.last_plan())
        assert np.array_equal(ra["logits"], rb["logits"]) and np.array_equal(ra["pred"], rb["pred"])
        tol_ok(ra["logits"][:256], orc.Oracle(sd).forward_windows(x[:256])["logits"], f"fp32_split (alias), {n} windows")
    g = a.split_guard()
    assert not g["enabled"] and not g["refused"] and "DCE_FP32_F16X2" in g["reason"], g
    a.close(); b.close()


def test_the_plan_table_is_what_runs(sd):
    """csrc/dce_api.hip kPlanRows (printed by tools/gen_options_table.py into DESIGN.md's appendix): for every precision of the product library, the kernel
    families dce_last_plan reports on both sides of every row boundary."""
    from conftest import has_experiments
    from deep_contact_estimator_amd import contact_cnn
    if has_experiments():
        pytest.skip("the experiments build plans with round 5's predicate tree")
    expect = {"fp32": {1: "conv_wino_quarter_ch2", 4096: "conv_wino2"},
              "bf16_fc": {256: "conv_x2_bf16_permk", 257: "conv_h2_bf16_permk"},
              "fp32_f16x2": {127: "conv_wino_half", 128: "conv_h2_f32", 1280: "conv_h2_f32", 1281: "conv_h2", 12288: "conv_h2", 12289: "conv_h2"}}
    fcs = {("fp32_f16x2", 1280): "fc_phased128x64", ("fp32_f16x2", 1281): "fc_h2_256x128_out2", ("fp32_f16x2", 12288): "fc23_fused_h2_128x64",
           ("fp32_f16x2", 12289): "fc_h2_256x128", ("fp32", 4096): "fc23_fused_phased128x64"}
    x = np.random.default_rng(1).standard_normal((12289, 150, 54), dtype=np.float32)
    for precision, rows in expect.items():
        m = contact_cnn(device=0, max_batch=16384, precision=precision); m.load_state_dict(sd).eval()
        for n, conv in rows.items():
            m.predict(x[:n])
            plan = m.last_plan()
            assert plan[0] == conv, (precision, n, plan)
            if (precision, n) in fcs:
                assert fcs[(precision, n)] in plan, (precision, n, plan)
        m.close()

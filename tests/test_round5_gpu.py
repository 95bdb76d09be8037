"""GPU (-m gpu), round 5: the range guard of the fp32_split precision (include/dce.h dce_split_guard_info; reference
utils/data_handler.py:55-56 is what bounds z-scored inputs, src/contact_cnn.py:10-58 what the static bounds are taken over), the
option string of dce_create_ex, the latency mode of the reference's shipped batch_size 1 (config/inference_one_seq_params.yaml:10)."""
import numpy as np
import pytest

from conftest import tol_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _model(precision, sd, max_batch=4096, tune=None):
    from deep_contact_estimator_amd import contact_cnn
    m = contact_cnn(device=0, max_batch=max_batch, precision=precision, tune=tune)
    m.load_state_dict(sd).eval()
    return m


def _argmax_contract(pred, ref):
    """argmax exact wherever the reference's top-2 margin exceeds 1e-3 of the largest logit."""
    srt = np.sort(ref["logits"], axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3 * np.abs(ref["logits"]).max()
    assert np.array_equal(pred[safe], ref["pred"][safe])


# ------------------------------------------------------------------------------------------------
# fp32_split: the static half of the guard
# ------------------------------------------------------------------------------------------------
@pytest.mark.experiments
def test_split_guard_static_bounds_of_a_checkpoint():
    """dce_finalize_weights bounds every layer's activations by sums of |w| (gain X + offs for |x| <= X) and derives the largest input
    the three-term split is safe for; the numbers are the ones numpy gets from the same checkpoint, and an ordinary checkpoint
    leaves room for any z-scored window (|z| <= 149 / sqrt(150))."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    m = _model("fp32_split", sd)
    g = m.split_guard()
    assert g["enabled"] and not g["refused"] and g["reason"] == "ok", g
    assert abs(g["z_max"] - 149 / np.sqrt(150)) < 1e-5
    gain, offs = 1.0, 0.0
    for l, (wk, bk) in enumerate((("block1.0.weight", "block1.0.bias"), ("block1.2.weight", "block1.2.bias"), ("block2.0.weight", "block2.0.bias"),
                                  ("block2.2.weight", "block2.2.bias"), ("fc.0.weight", "fc.0.bias"), ("fc.3.weight", "fc.3.bias"))):
        w = np.abs(sd[wk].astype(np.float64)).reshape(sd[wk].shape[0], -1)
        s, b = w.sum(1).max(), np.abs(sd[bk].astype(np.float64)).max()
        gain, offs = gain * s, offs * s + b
        assert np.isclose(g["gain"][l], gain, rtol=1e-9) and np.isclose(g["offs"][l], offs, rtol=1e-9), (l, g["gain"][l], gain)
    x_hi = min((2.0 ** 126 - g["offs"][l]) / g["gain"][l] for l in range(4))      # conv1..4 outputs: the activations that are split (h1 too with x3_fc3=1)
    assert g["x_hi"] <= x_hi and g["x_hi"] >= x_hi * (1 - 1e-6) and g["x_hi"] > 1e20 and g["x_lo"] == 2.0 ** -40, g
    # a model of another precision reports the guard as not applicable
    f = _model("fp32", sd); gf = f.split_guard()
    assert not gf["enabled"] and not gf["refused"]
    m.close(); f.close()


@pytest.mark.experiments
@pytest.mark.parametrize("case", ["tiny_conv1", "huge_fc0", "nan_weight", "huge_gain"])
def test_split_guard_refuses_a_checkpoint_outside_the_range(case, orc):
    """A checkpoint whose weights leave the range in which a three-term split is exact -- a layer whose largest |w| is below 2^-40
    (the audit's 'top binade' set: conv1 x 3.3e-39), at bf16's range limit, non-finite, or whose activation bounds admit no z-scored
    window -- is refused at dce_finalize_weights: the model then runs the DCE_FP32 kernels for every call (plan says so) and returns
    the fp32 precision's bits."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    if case == "tiny_conv1":
        sd["block1.0.weight"] = (sd["block1.0.weight"] * np.float32(1e-38 / 3)).astype(np.float32)
    elif case == "huge_fc0":
        sd["fc.0.weight"][3, 7] = np.float32(3.0e38)
    elif case == "nan_weight":
        sd["block2.0.weight"][5, 5, 1] = np.nan
    else:                                                         # every layer x 1e9: the bound of conv4 passes 2^126 for |z| <= 12.2
        for k in ("block1.0.weight", "block1.2.weight", "block2.0.weight", "block2.2.weight"):
            sd[k] = (sd[k] * np.float32(1e9)).astype(np.float32)
    a, b = _model("fp32_split", sd), _model("fp32", sd)
    g = a.split_guard()
    assert g["enabled"] and g["refused"] and g["reason"] != "ok", g
    x = np.random.default_rng(3).standard_normal((3000, 150, 54), dtype=np.float32)
    seq = np.random.default_rng(4).standard_normal((3000 + 149, 54)).astype(np.float32)
    for fa, fb, inp in ((a.predict, b.predict, x), (a.infer_sequence, b.infer_sequence, seq)):
        ra = fa(inp); plan = a.last_plan()
        rb = fb(inp)
        assert plan[0] == "split_guard_refused" and not any(k.startswith(("conv_x3", "fc_x3")) for k in plan), plan
        assert np.array_equal(ra["logits"].view(np.uint32), rb["logits"].view(np.uint32)) and np.array_equal(ra["pred"], rb["pred"])
    a.close(); b.close()


# ------------------------------------------------------------------------------------------------
# fp32_split: the dynamic half (pre-normalised windows), and the gated fp32 fallback behind it
# ------------------------------------------------------------------------------------------------
def _adversarial(kind, n, rng):
    """(state_dict, windows) of tools/precision_audit.py's pre-normalised sets, with weights that PASS the static guard."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    win = rng.standard_normal((n, 150, 54)).astype(np.float32)
    if kind == "top_binade":                                      # |x| up to 1.6e38, above the guard's 2^126 (a first term rounds to Inf from 3.39e38 on; the fp32
        win *= np.float32(3.0e37)                                  # Winograd path's own input transform d_i +- d_j needs |x| < 1.7e38); conv1 x 1e-8 keeps the net finite
        idx = rng.integers(0, win.size, 50 * max(n // 256, 1))
        win.reshape(-1)[idx] = np.float32(1.6e38) * np.sign(win.reshape(-1)[idx])
        sd["block1.0.weight"] = (sd["block1.0.weight"] * np.float32(1e-8)).astype(np.float32)
    elif kind == "subnormal_terms":                               # inputs x 2^-100 (third terms of their split are subnormal), conv2 x 2^100 brings the net back
        win *= np.float32(2.0 ** -100)
        sd["block1.0.bias"] = (sd["block1.0.bias"] * np.float32(2.0 ** -100)).astype(np.float32)
        sd["block1.2.weight"] = (sd["block1.2.weight"] * np.float32(2.0 ** 100)).astype(np.float32)
    elif kind == "one_window":                                    # ONE window of the launch above x_hi; ordinary weights (its logits overflow in every evaluation)
        win[n // 2] *= np.float32(1e37)
    return sd, win


@pytest.mark.parametrize("n", [300, 4096])                        # conv_x3_f32 + fp32 FC kernels / conv_x3_permk + fc_x3
@pytest.mark.experiments
@pytest.mark.parametrize("kind", ["top_binade", "subnormal_terms", "one_window"])
def test_split_guard_routes_out_of_range_windows_to_the_fp32_kernels(kind, n, orc):
    """Pre-normalised windows outside [x_lo, x_hi] -- where the first term of a split would round to Inf, or the third terms go
    subnormal (profiles/r4_precision_audit.json: 4713 bounds / 0.94 of the bound without a guard) -- are seen by the conv kernel's load
    stage; the launch is then recomputed by the gated DCE_FP32 sequence queued behind it: the results are the fp32 precision's bits,
    within the contract of the oracle wherever the oracle is finite, for host and device callers alike (nothing comes back to the host)."""
    import torch
    rng = np.random.default_rng(11 + n)
    sd, win = _adversarial(kind, n, rng)
    a, b = _model("fp32_split", sd), _model("fp32", sd)
    g0 = a.split_guard()
    assert g0["enabled"] and not g0["refused"], g0
    ra, rb = a.predict(win), b.predict(win)
    assert a.last_plan()[0].startswith("conv_x3") and a.last_plan()[-1] == "gated_fp32_fallback", a.last_plan()
    g1 = a.split_guard()
    assert g1["guarded_launches"] == g0["guarded_launches"] + 1 and g1["fallbacks_run"] == g0["fallbacks_run"] + 1, (g0, g1)
    assert g1["windows_out_of_range"] - g0["windows_out_of_range"] == (1 if kind == "one_window" else n), g1
    assert np.array_equal(ra["logits"].view(np.uint32), rb["logits"].view(np.uint32)), kind
    assert np.array_equal(ra["pred"], rb["pred"]) and np.array_equal(ra["contacts"], rb["contacts"])
    ref = orc.Oracle(sd).forward_windows(win)
    fin = np.isfinite(ref["logits"]).all(1)
    assert fin.sum() >= n - 1
    tol_ok(ra["logits"][fin], ref["logits"][fin], f"{kind}: guarded fp32_split vs oracle")
    _argmax_contract(ra["pred"][fin], {"logits": ref["logits"][fin], "pred": ref["pred"][fin]})
    # device pointers, asynchronous on torch's stream: same bits, and the next (in-range) launch runs the split kernels alone
    xt = torch.from_numpy(win).cuda()
    rt = a.predict(xt)
    assert np.array_equal(rt["logits"].cpu().numpy().view(np.uint32), rb["logits"].view(np.uint32))
    ok = rng.standard_normal((n, 150, 54)).astype(np.float32)
    ok[1] = 0.0                                                    # an all-zero window is inside the range (zeros split exactly)
    ro = a.predict(ok)
    g2 = a.split_guard()
    assert g2["fallbacks_run"] == g1["fallbacks_run"] + 1 and g2["guarded_launches"] == g1["guarded_launches"] + 2, (g1, g2)   # (+1: the device-pointer call above)
    assert g2["windows_out_of_range"] == g1["windows_out_of_range"] + (1 if kind == "one_window" else n)
    if kind == "one_window":                                       # ordinary weights: the in-range launch is an ordinary fp32_split launch
        tol_ok(ro["logits"], orc.Oracle(sd).forward_windows(ok)["logits"], "in-range launch after a fallback")
        assert not np.array_equal(ro["logits"], b.predict(ok)["logits"])          # ... on the split kernels (another association of the sums)
    a.close(); b.close()


@pytest.mark.experiments
def test_split_guard_leaves_the_zscore_entry_alone_and_can_be_switched_off(orc):
    """z-scored windows are inside the range by construction: dce_infer_sequence carries no per-window check and no gated sequence;
    split_guard=0 restores the round-4 behaviour (A/B, the audit's 'no guard' rows)."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    seq = synth.make_sequence(4096 + 149, seed=2).astype(np.float32)
    a = _model("fp32_split", sd)
    out = a.infer_sequence(seq)
    assert "gated_fp32_fallback" not in a.last_plan() and a.last_plan()[0].startswith("conv_x3"), a.last_plan()
    assert a.split_guard()["guarded_launches"] == 0
    ref = orc.Oracle(sd).infer_sequence(seq)
    tol_ok(out["logits"], ref["logits"], "fp32_split, z-score entry")
    off = _model("fp32_split", sd, tune={"split_guard": 0})
    x = np.random.default_rng(5).standard_normal((4096, 150, 54), dtype=np.float32)
    r_off, r_on = off.predict(x), a.predict(x)
    assert "gated_fp32_fallback" not in off.last_plan() and a.last_plan()[-1] == "gated_fp32_fallback"
    assert not off.split_guard()["enabled"]
    assert np.array_equal(r_off["logits"], r_on["logits"])        # in range: the guard changes nothing
    a.close(); off.close()


@pytest.mark.experiments
def test_bf16_gemm_k32_two_workgroups_per_cu_bit_identical():
    """Round 5's one experiment on the bf16 fc.0 GEMM (option bf16_k32=1, experiments build): the 256 x 128 tile with 32-k K-tiles, two
    workgroups per CU.  Same MFMA sequence per output as the shipped 64-k kernel: the same bits, a partial last row tile and repeated
    runs included (measured 22 % slower: profiles/r5i_gemm_k32.txt -- it stays an experiment)."""
    import torch
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    for n in (8192, 16384 - 100):                                  # (sizes whose rounds model picks the 256 x 128 tile; the second with a partial last row tile)
        x = torch.randn((n, 150, 54), generator=torch.Generator(device="cuda").manual_seed(n), device="cuda")
        a = contact_cnn(device=0, max_batch=n, precision="bf16_fc", tune={"bf16_k32": 1}); a.load_state_dict(sd).eval()
        b = contact_cnn(device=0, max_batch=n, precision="bf16_fc", tune={"bf16_k32": 0}); b.load_state_dict(sd).eval()
        rb = b.predict(x)["logits"].clone()
        for rep in range(3):
            ra = a.predict(x)["logits"]
            assert "fc_phased256x128_k32_bf16" in a.last_plan() and "fc_phased256x128_k32_bf16" not in b.last_plan(), (a.last_plan(), b.last_plan())
            assert torch.equal(ra, rb), (n, rep)
        a.close(); b.close()


# ------------------------------------------------------------------------------------------------
# latency mode (csrc/latency.hip): one window in one kernel; online pushes through a resident kernel and a mailbox
# ------------------------------------------------------------------------------------------------
def test_latency_mode_one_window_calls_vs_oracle(orc):
    """option latency=1: a one-window call (the reference's shipped batch_size 1, config/inference_one_seq_params.yaml:10) is ONE
    kernel -- four conv-segment workgroups and 252 fc workgroups that hold their neurons' weights in registers, meeting through
    arrival counters.  Not the batch path's bits (K is folded over the lanes by a tree), but inside the fp32 contract of the ORACLE:
    pre-normalised and raw (z-score fused) windows, host and device pointers, packed rows, a non-finite window; two windows and
    more take the batch path's kernels."""
    import torch
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    m, b = _model("fp32", sd, max_batch=64, tune={"latency": 1}), _model("fp32", sd, max_batch=64)
    seq = synth.make_sequence(150 + 95, seed=21, kind="ar1").astype(np.float32)
    zw = orc.zscore_windows(seq)
    ref = orc.Oracle(sd).forward_windows(zw)
    n = zw.shape[0]
    got = np.stack([m.predict(zw[i:i + 1])["logits"][0] for i in range(n)])
    assert m.last_plan() == ["latency_one"], m.last_plan()
    tol_ok(got, ref["logits"], "latency mode, one pre-normalised window per call")
    pred = np.array([int(m.predict(zw[i:i + 1])["pred"][0]) for i in range(n)])
    _argmax_contract(pred, ref)
    # raw rows: dce_infer_sequence with T = 150
    got_z = np.stack([m.infer_sequence(seq[i:i + 150])["logits"][0] for i in range(n)])
    assert m.last_plan() == ["latency_one_zs"], m.last_plan()
    tol_ok(got_z, ref["logits"], "latency mode, one raw window per call")
    # device pointers (asynchronous on torch's stream), contacts, packed rows; repeated calls are bit-stable
    xt = torch.from_numpy(zw).cuda()
    outs = [m.predict(xt[i:i + 1]) for i in range(n)]
    torch.cuda.synchronize()
    dev = np.stack([o["logits"][0].cpu().numpy() for o in outs])
    assert np.array_equal(dev, got)
    assert np.array_equal(np.stack([o["contacts"][0].cpu().numpy() for o in outs]), orc.decimal2binary(pred))
    pk = m.predict_packed(xt[3:4]).cpu().numpy()
    assert np.array_equal(pk[0, :64].view(np.float32), got[3]) and np.array_equal(pk[0, 64:], orc.decimal2binary(pred[3:4])[0])
    # the batch path is within the tolerance of it too, and is what two or more windows run
    tol_ok(got, b.predict(zw)["logits"], "latency mode vs the batch path")
    two = m.predict(zw[:2])                                         # (round 6: two .. 32 windows have a one-kernel form of their own, tests/test_round6_gpu.py)
    assert m.last_plan() == ["latency_mb"], m.last_plan()
    tol_ok(two["logits"], ref["logits"][:2], "latency mode, two windows")
    many = m.predict(zw[:40])
    assert not any(k.startswith("latency") for k in m.last_plan()) and np.array_equal(many["logits"], b.predict(zw[:40])["logits"])
    bad = zw[5:6].copy(); bad[0, 17, 3] = np.nan
    r = m.predict(bad)
    assert np.isnan(r["logits"]).all() and r["pred"][0] == 0
    tol_ok(m.predict(zw[6:7])["logits"], ref["logits"][6:7], "after a non-finite window")
    m.close(); b.close()


def test_latency_mode_online_pushes_vs_oracle(orc):
    """dce_online_push in the latency mode: a resident kernel takes each sample from a mailbox in pinned memory (no launch, no copy
    per push), keeps the last 150 in LDS and answers through the mailbox.  4,350 pushes -- beyond the 4096-row compaction of the
    batch path's sample buffer -- every estimate against the ORACLE on the same rows (fp32 contract + argmax), then: the kernel leaves
    by itself when no sample comes (latency_idle_ms) and the next push restarts it with the history intact; another call on the model
    (predict) makes it leave and pushes continue; a reset starts a new sequence."""
    import time
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    T = 150 + 4200
    seq = synth.make_sequence(T, seed=33).astype(np.float32)
    ref = orc.Oracle(sd).infer_sequence(seq)
    m = _model("fp32", sd, max_batch=64, tune={"latency": 1, "latency_idle_ms": 30})
    m.online_reset()
    rows = []
    for t in range(T):
        if t == 2000:
            time.sleep(0.2)                                        # the service leaves (30 ms without a sample) ...
        if t == 3000:
            m.predict(seq[:150][None] * 0 + 1)                     # ... or is told to: another call needs the device
        r = m.online_push(seq[t])
        assert (r is None) == (t < 149), t
        if r is not None:
            rows.append((r[0].copy(), r[1], r[2].copy()))
    logits = np.stack([r[0] for r in rows]); pred = np.array([r[1] for r in rows]); contacts = np.stack([r[2] for r in rows])
    assert logits.shape == ref["logits"].shape
    tol_ok(logits, ref["logits"], "latency mode, online pushes vs the oracle")
    _argmax_contract(pred, ref)
    assert np.array_equal(contacts, orc.decimal2binary(pred))
    assert np.array_equal(pred, logits.argmax(1))
    # a reset starts over: the first 149 pushes answer nothing, the 150th is window 0 of the new sequence
    m.online_reset()
    seq2 = synth.make_sequence(150 + 20, seed=34, kind="ar1").astype(np.float32)
    ref2 = orc.Oracle(sd).infer_sequence(seq2)
    got2 = [m.online_push(s) for s in seq2]
    assert all(g is None for g in got2[:149])
    tol_ok(np.stack([g[0] for g in got2[149:]]), ref2["logits"], "after a reset")
    m.close()


# ------------------------------------------------------------------------------------------------
# fp32_split: fc.3 on three-term operands (csrc/fc_gemm_x3.hip, the 128 x 64 tile with the fused fc.6 epilogue)
# ------------------------------------------------------------------------------------------------
@pytest.mark.experiments
@pytest.mark.parametrize("n", [4096, 4100, 8192])
def test_split_mode_fc3_on_three_term_operands(n, orc):
    """Option x3_fc3=1: fp32_split at chip-filling batches with fc.3 on the bf16 matrix pipe too -- fc.0's epilogue writes h1 as three bf16
    planes, the 128 x 64 tile of fc_gemm_x3 multiplies them with fc.3's three-plane weights and finishes fc.6's chunk sums in its epilogue
    (plan fc23_fused_x3_128x64).  Same contract as the fp32 path against the oracle -- every row --, within 2e-5 of the largest logit of
    the default form of the mode (fc.3 on fp32 MFMA), h2 taps within the tolerance, a non-finite window contained, repeatable.  (Measured
    no faster than the fp32 kernel -- profiles/r5j_split_fc3.txt -- hence an option, not the default.)"""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model("fp32_split", sd, max_batch=8192, tune={"x3_fc3": 1})
    b = _model("fp32_split", sd, max_batch=8192)
    x = np.random.default_rng(40 + n).standard_normal((n, 150, 54), dtype=np.float32)
    x[11, 5, 7] = np.inf
    ra, rb = a.predict(x), b.predict(x)
    assert "fc23_fused_x3_128x64" in a.last_plan() and "fc_x3_256x128" in a.last_plan(), a.last_plan()
    assert "fc23_fused_x3_128x64" not in b.last_plan() and "fc23_fused_phased128x64" in b.last_plan(), b.last_plan()
    assert np.array_equal(a.predict(x)["logits"].view(np.uint32), ra["logits"].view(np.uint32))
    ok = np.ones(n, bool); ok[11] = False
    assert np.isnan(ra["logits"][11]).all() and ra["pred"][11] == 0 and np.isfinite(ra["logits"][ok]).all()
    ref = orc.Oracle(sd).forward_windows(x[ok])
    tol_ok(ra["logits"][ok], ref["logits"], f"fp32_split with fc.3 on three-term operands, {n} windows")
    _argmax_contract(ra["pred"][ok], ref)
    scale = np.abs(ref["logits"]).max()
    assert np.abs(ra["logits"][ok] - rb["logits"][ok]).max() <= 2e-5 * scale
    if n == 4096:
        t = a.forward_taps(x[:4096])                                   # h1 wanted in fp32: that call keeps fc.3 on the fp32 kernels
        assert "fc23_fused_x3_128x64" not in a.last_plan()
        rt = orc.Oracle(sd).forward_windows(x[ok][:64], taps=True)
        tol_ok(t["h2"][ok[:4096]][:64], rt["h2"], "h2 tap")
    a.close(); b.close()

"""GPU (-m gpu): the DCE_FP32_F16X2 precision ("fp32_f16x2") -- the conv stack and fc.0 of reference src/contact_cnn.py:10-49 at chip-filling
batches on the fp16 matrix pipe, every operand as two fp16 terms of the value times a power of two, three MFMAs per product
(csrc/conv_h2.hip, csrc/fc_gemm_h2.hip).

The mode claims the fp32 TOLERANCE, so it is held to the same contract as the fp32 path -- |got - ref| <= 1e-5 max|ref| + 1e-4 |ref| against the
oracle (fp64 accumulation of the reference's arithmetic), argmax exact outside the noise margin -- and, because its scales are chosen per window
inside the kernel, to that contract over the whole fp32 range of inputs and checkpoints: there is no guard to trip and no fallback to take.
"""
import numpy as np
import pytest

from conftest import tol_ok

pytestmark = pytest.mark.gpu
N0 = 3072                                                   # a chip-filling launch (the two-term FC kernels run from 1281 windows: 96 tiles of 256 x 128)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def pair():
    """(fp32 model, fp32_f16x2 model) on the bench checkpoint."""
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    a = contact_cnn(device=0, max_batch=8192); a.load_state_dict(sd).eval()
    b = contact_cnn(device=0, max_batch=8192, precision="fp32_f16x2"); b.load_state_dict(sd).eval()
    yield sd, a, b
    a.close(); b.close()


def _model(sd, max_batch=8192, tune=None):
    from deep_contact_estimator_amd import contact_cnn
    m = contact_cnn(device=0, max_batch=max_batch, precision="fp32_f16x2", tune=tune)
    m.load_state_dict(sd).eval()
    return m


def _argmax_ok(got_pred, ref_logits, ref_pred):
    srt = np.sort(ref_logits, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3 * np.abs(ref_logits).max()
    assert np.array_equal(got_pred[safe], ref_pred[safe])
    flips = int((got_pred != ref_pred).sum())
    assert flips <= max(1, int(2e-3 * len(ref_pred))), flips
    return flips


def _is_h2(plan):
    return plan[0] == "conv_h2" and any(k in plan for k in ("fc_h2_256x128", "fc_h2_256x128_out2"))


@pytest.mark.parametrize("n", [1281, 2049, 3072, 4096, 4100, 8192])
def test_mode_meets_the_fp32_contract(n, pair, orc):
    """Every row of a chip-filling batch against the oracle at the tolerance of the fp32 path, pre-normalised windows and the z-score entry
    (4100 = a ragged last tile); and in the error class of the fp32 MFMA path, not of a 16-bit one."""
    from deep_contact_estimator_amd import synth
    sd, a, b = pair
    seq = synth.make_sequence(n + 149, seed=2).astype(np.float32)
    w = orc.zscore_windows(seq)
    ref = orc.Oracle(sd).forward_windows(w)
    scale = np.abs(ref["logits"]).max()
    for what, out in (("windows", b.predict(w)), ("sequence", b.infer_sequence(seq))):
        assert _is_h2(b.last_plan()), b.last_plan()
        tol_ok(out["logits"], ref["logits"], f"fp32_f16x2 {what}, {n} rows vs oracle")
        _argmax_ok(out["pred"], ref["logits"], ref["pred"])
        assert np.array_equal(out["contacts"], orc.decimal2binary(out["pred"]))
        nat = a.predict(w)
        e_h2 = np.abs(out["logits"].astype(np.float64) - ref["logits"]).max()
        e_nat = np.abs(nat["logits"].astype(np.float64) - ref["logits"]).max()
        print(f"n={n} {what}: max|f16x2 - oracle| {e_h2:.2e}, max|fp32 - oracle| {e_nat:.2e}, scale {scale:.2f}")
        assert np.abs(out["logits"].astype(np.float64) - nat["logits"]).max() < 2e-5 * scale
        assert e_h2 < 4 * max(e_nat, 1e-6 * scale)


@pytest.mark.parametrize("case", ["seq_normal", "seq_ar1"])
def test_conv_h2_layers_vs_reference_hooks(case, golden, case_inputs, orc):
    """The layers INSIDE the two-term fp16 conv stack against the reference's own forward-hook taps (tests/golden/make_golden.py), and the other
    windows of the launch against the oracle: conv_h2_kernel's TAPS instantiation = the product kernel plus stores."""
    g = golden(case)
    sd, seq = case_inputs(g)
    m = _model(sd, 64)
    zw = orc.zscore_windows(seq)[:64]
    t = m.conv_layer_taps(zw, "h2")
    o = orc.Oracle(sd)
    for k in ("conv1", "conv2", "pool1", "conv3", "conv4"):
        if "tap_" + k in g.files:
            tol_ok(t[k][0], g["tap_" + k], f"{case}: {k} of window 0 vs the reference's hook")
    lt = [o.layer_taps(w) for w in zw[:8]]
    for k in ("conv1", "conv2", "pool1", "conv3", "conv4"):
        tol_ok(t[k][:8], np.stack([x[k] for x in lt]), f"{case}: {k}, 8 windows vs oracle")
    tol_ok(t["feat"], o.forward_windows(zw, taps=True)["feat"], f"{case}: features, 64 windows vs oracle")
    m.close()


@pytest.mark.parametrize("n", [128, 700, 1280])
def test_mid_size_batches(n, pair, orc):
    """From 128 windows up to fc_gemm_h2's threshold the mode's conv stack already runs on two-term fp16 operands (conv_h2_kernel with fp32
    features out; the FC layers stay on the fp32 kernels): every row against the oracle at the fp32 tolerance, a NaN window contained; and
    the features of a chip-filling tap are the same kernel's."""
    sd, a, b = pair
    x = np.random.default_rng(1000 + n).standard_normal((n, 150, 54), dtype=np.float32)
    x[3, 100, 0] = np.nan
    out = b.predict(x)
    plan = b.last_plan()
    assert plan[0] == "conv_h2_f32" and "fc_h2_256x128" not in plan, plan
    assert np.isnan(out["logits"][3]).all() and out["pred"][3] == 0
    keep = np.arange(n) != 3
    o = orc.Oracle(sd)
    ref = o.forward_windows(x[keep])
    tol_ok(out["logits"][keep], ref["logits"], f"fp32_f16x2, mid-size batch {n}")
    _argmax_ok(out["pred"][keep], ref["logits"], ref["pred"])
    t = b.forward_taps(x[:256] if n >= 256 else x)
    assert b.last_plan()[0] == "conv_h2_f32", b.last_plan()
    k2 = keep[:len(t["feat"])]
    tol_ok(t["feat"][k2], o.forward_windows(x[:len(t["feat"])][k2], taps=True)["feat"], "features of the two-term fp16 conv stack")


@pytest.mark.parametrize("n", [1, 30, 127])
def test_small_batches_are_the_fp32_path(n, pair):
    """Below 128 windows the mode runs the DCE_FP32 kernels: that precision's bits."""
    sd, a, b = pair
    x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
    ra, rb = a.predict(x), b.predict(x)
    assert not any(k.startswith(("conv_h2", "fc_h2")) for k in b.last_plan()), b.last_plan()
    for k in ("logits", "pred", "contacts"):
        assert np.array_equal(ra[k], rb[k]), k


def test_a_window_depends_on_that_window_alone(pair):
    """The scales are per window: the same window returns the same bits whatever shares its launch, wherever it sits in it, and through
    both entries' load stages' common arithmetic (pre-normalised windows)."""
    sd, a, b = pair
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N0, 150, 54), dtype=np.float32)
    base = b.predict(x)["logits"]
    y = np.concatenate([rng.standard_normal((1000, 150, 54), dtype=np.float32) * 1e6, x[::-1], rng.standard_normal((100, 150, 54), dtype=np.float32) * 1e-6])
    out = b.predict(y)
    assert _is_h2(b.last_plan())
    assert np.array_equal(out["logits"][1000:1000 + N0][::-1], base)


def test_non_finite_and_degenerate_windows(pair, orc):
    """NaN / Inf samples: NaN logits, class 0, contained in their window; an all-zero window and a window of fp32 subnormals: the bias path."""
    sd, a, b = pair
    x = np.random.default_rng(8).standard_normal((N0, 150, 54), dtype=np.float32)
    x[3, 100, 0] = np.nan
    x[500, 0, 53] = np.inf
    x[501, 149, 7] = -np.inf
    x[700] = 0.0
    x[701] = np.float32(1e-42)
    x[702, :, :27] = 0.0
    out = b.predict(x)
    assert _is_h2(b.last_plan())
    for r in (3, 500, 501):
        assert np.isnan(out["logits"][r]).all() and out["pred"][r] == 0
    keep = np.ones(N0, bool); keep[[3, 500, 501]] = False
    ref = orc.Oracle(sd).forward_windows(x[keep])
    tol_ok(out["logits"][keep], ref["logits"], "rows beside the non-finite ones, the zero / subnormal windows among them")
    _argmax_ok(out["pred"][keep], ref["logits"], ref["pred"])
    # the z-score entry: a constant channel has no standard deviation -> NaN window -> class 0 (utils/data_handler.py:55-56)
    from deep_contact_estimator_amd import synth
    seq = synth.make_sequence(N0 + 149, seed=4).astype(np.float32)
    seq[1000:1200, 5] = 2.5
    o = b.infer_sequence(seq)
    refz = orc.Oracle(sd).infer_sequence(seq)
    bad = np.isnan(refz["logits"]).any(1)
    assert bad.sum() >= 51 and np.array_equal(np.isnan(o["logits"]).any(1), bad) and (o["pred"][bad] == 0).all()
    tol_ok(o["logits"][~bad], refz["logits"][~bad], "z-score entry beside the constant-channel windows")


@pytest.mark.parametrize("e", [-120, -60, -20, 20, 60, 100])
def test_input_scale_sweep(e, orc):
    """Pre-normalised windows times 2^e, conv1's weights times 2^-e (the rest of the net sees ordinary numbers): 2^-120 .. 2^100 -- far outside
    fp16's own range, and outside the range DCE_FP32_SPLIT's guard admits -- within the contract, same plan, no fallback."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    sd["block1.0.weight"] = (sd["block1.0.weight"] * np.float32(2.0 ** -e)).astype(np.float32)
    x = (np.random.default_rng(400 + e).standard_normal((N0, 150, 54), dtype=np.float32) * np.float32(2.0 ** e)).astype(np.float32)
    m = _model(sd, N0)
    out = m.predict(x)
    assert _is_h2(m.last_plan()), m.last_plan()
    rows = np.r_[0:128, N0 - 128:N0]
    ref = orc.Oracle(sd).forward_windows(x[rows])
    assert np.isfinite(ref["logits"]).all()
    tol_ok(out["logits"][rows], ref["logits"], f"inputs x 2^{e}")
    _argmax_ok(out["pred"][rows], ref["logits"], ref["pred"])
    m.close()


@pytest.mark.parametrize("e", [-100, -118])
def test_audit_set_inputs_and_conv1_bias_scaled_down_conv2_weights_up(e, orc):
    """tools/precision_audit.py's `inputs_x_2^e` sets: windows x 2^e, conv1's bias with them, conv2's weights x 2^-e -- conv1's activations sit at
    2^e, conv2's weights at 2^-e: the scale exponents of the input (14 - e) and of conv2's weights leave any 'comfortable' range (the first
    build capped S + sw at 100 and lost these sets by four orders of magnitude: profiles/r5p_precision_audit.json)."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    sd["block1.0.bias"] = (sd["block1.0.bias"] * np.float32(2.0 ** e)).astype(np.float32)
    sd["block1.2.weight"] = (sd["block1.2.weight"] * np.float32(2.0 ** -e)).astype(np.float32)
    x = (np.random.default_rng(2026).standard_normal((N0, 150, 54), dtype=np.float32) * np.float32(2.0 ** e)).astype(np.float32)
    m = _model(sd, N0)
    out = m.predict(x)
    assert _is_h2(m.last_plan()), m.last_plan()
    rows = np.r_[0:128, N0 - 128:N0]
    ref = orc.Oracle(sd).forward_windows(x[rows])
    assert np.isfinite(ref["logits"]).all()
    tol_ok(out["logits"][rows], ref["logits"], f"audit set inputs x 2^{e}")
    _argmax_ok(out["pred"][rows], ref["logits"], ref["pred"])
    m.close()


def test_audit_set_inputs_in_the_top_binade(orc):
    """tools/precision_audit.py's `inputs_in_the_top_binade`: windows x 3e37 with samples at +-3.395e38 (above bf16's largest finite number, 2^128
    within a hair), conv1's weights x 3.3e-39 -- fp32 SUBNORMALS, which the host scales up exactly (sw = 14 + 131)."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    sd["block1.0.weight"] = (sd["block1.0.weight"] * 1e-38 / 3.0).astype(np.float32)
    rng = np.random.default_rng(2027)
    x = (rng.standard_normal((N0, 150, 54), dtype=np.float32) * np.float32(3.0e37)).astype(np.float32)
    idx = rng.integers(0, x.size, 2000)
    x.reshape(-1)[idx] = np.float32(3.395e38) * np.sign(x.reshape(-1)[idx])
    m = _model(sd, N0)
    out = m.predict(x)
    assert _is_h2(m.last_plan()), m.last_plan()
    rows = np.r_[0:128, N0 - 128:N0]
    ref = orc.Oracle(sd).forward_windows(x[rows])
    assert np.isfinite(ref["logits"]).all()
    tol_ok(out["logits"][rows], ref["logits"], "audit set: top binade")
    _argmax_ok(out["pred"][rows], ref["logits"], ref["pred"])
    m.close()


@pytest.mark.parametrize("kind", ["alternating_1e3", "alternating_1e-3", "tiny_weights_large_bias", "large_bias_everywhere", "zero_bias", "mixed_window_scales"])
def test_checkpoint_and_window_scale_sweep(kind, orc):
    """Checkpoints whose layers swing the activations over many decades, biases far above / below the products, and windows of very different
    magnitude inside one launch (each gets its own scales)."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "zero" if kind == "zero_bias" else "uniform").items()}
    layers = (("block1.0", 1), ("block1.2", 1), ("block2.0", 1), ("block2.2", 1), ("fc.0", 1), ("fc.3", 1))
    rng = np.random.default_rng(17)
    x = rng.standard_normal((N0, 150, 54), dtype=np.float32)
    if kind.startswith("alternating"):
        f0 = 1e3 if kind.endswith("1e3") else 1e-3
        cum = 1.0
        for i, (name, _) in enumerate(layers):
            f = f0 if i % 2 == 0 else 1.0 / f0
            cum *= f
            sd[name + ".weight"] = (sd[name + ".weight"] * np.float32(f)).astype(np.float32)
            sd[name + ".bias"] = (sd[name + ".bias"] * np.float32(cum)).astype(np.float32)
    elif kind == "tiny_weights_large_bias":
        sd["block1.2.weight"] = (sd["block1.2.weight"] * np.float32(1e-12)).astype(np.float32)      # conv2's products vanish beside its bias
        sd["block2.2.bias"] = (sd["block2.2.bias"] * np.float32(50.0)).astype(np.float32)
    elif kind == "large_bias_everywhere":
        for name, _ in layers[:4]:
            sd[name + ".bias"] = (sd[name + ".bias"] * np.float32(1e4)).astype(np.float32)
    elif kind == "mixed_window_scales":
        x *= (10.0 ** rng.uniform(-8, 8, N0)).astype(np.float32)[:, None, None]
    m = _model(sd, N0)
    out = m.predict(x)
    assert _is_h2(m.last_plan()), m.last_plan()
    rows = np.r_[0:160, N0 - 96:N0]
    ref = orc.Oracle(sd).forward_windows(x[rows])
    assert np.isfinite(ref["logits"]).all()
    tol_ok(out["logits"][rows], ref["logits"], kind)
    _argmax_ok(out["pred"][rows], ref["logits"], ref["pred"])
    m.close()


def test_non_finite_checkpoint_runs_the_fp32_kernels(pair):
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    sd["block2.0.weight"][5, 3, 1] = np.inf
    m = _model(sd, N0)
    a = contact_cnn(device=0, max_batch=N0); a.load_state_dict(sd).eval()
    x = np.random.default_rng(2).standard_normal((N0, 150, 54), dtype=np.float32)
    o, r = m.predict(x), a.predict(x)
    assert m.last_plan()[0] == "f16x2_refused" and not any(k.startswith(("conv_h2", "fc_h2")) for k in m.last_plan()), m.last_plan()
    assert np.array_equal(o["logits"], r["logits"], equal_nan=True) and np.array_equal(o["pred"], r["pred"])
    m.close(); a.close()


@pytest.mark.parametrize("n", [3072, 4096, 4100, 8192])
def test_fc3_on_two_term_operands_and_without(n, pair, orc):
    """fc.3 + fc.6's chunk sums on two-term fp16 operands (the default: h1 leaves fc.0 as two fp16 terms with a row scale from a Cauchy-Schwarz
    bound, fc_gemm_h2k_kernel<H2KFc3> with the fused fc.6 epilogue) against the option h2_fc3=0 (fp32 h1, the DCE_FP32 kernels behind fc.0):
    both within the contract of the oracle, fp32 rounding apart; h2 and h1 taps take the fp32 route."""
    sd, a, b = pair
    m0 = _model(sd, 8192, tune={"h2_fc3": 0})
    x = np.random.default_rng(900 + n).standard_normal((n, 150, 54), dtype=np.float32)
    x[n - 1] *= np.float32(1e-20)                              # a ragged tile's last row, with a scale of its own
    x[5] *= np.float32(1e15)
    o1, o0 = b.predict(x), m0.predict(x)
    assert "fc23_fused_h2_128x64" in b.last_plan() and "fc_h2_256x128_out2" in b.last_plan(), b.last_plan()
    assert "fc23_fused_h2_128x64" not in m0.last_plan() and "fc_h2_256x128" in m0.last_plan(), m0.last_plan()
    rows = np.r_[0:160, n - 96:n]
    ref = orc.Oracle(sd).forward_windows(x[rows])
    for what, o in (("fc.3 on two-term operands", o1), ("fc.3 on the fp32 kernels", o0)):
        tol_ok(o["logits"][rows], ref["logits"], f"{what}, {n} windows")
        _argmax_ok(o["pred"][rows], ref["logits"], ref["pred"])
    assert np.abs(o1["logits"].astype(np.float64) - o0["logits"]).max() < 2e-5 * np.abs(ref["logits"]).max()
    t = b.forward_taps(x[:3072])                             # taps: fp32 features / h1 / h2 -> the fp32 route behind conv_h2_f32
    assert "fc23_fused_h2_128x64" not in b.last_plan(), b.last_plan()
    tol_ok(t["logits"][:160], ref["logits"][:160], "taps route")
    m0.close()


def test_fc3_in_launches_past_the_fused_tile(orc):
    """32768 windows per launch (what the 1e6-window sequence runs): fc.3 on fc.0's 256 x 128 two-term kernel (h2 in fp32) + the tail kernel."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    m = _model(sd, 32768)
    n = 32768 + 5000
    seq = synth.make_sequence(n + 149, seed=12).astype(np.float32)
    out = m.infer_sequence(seq[:32768 + 149])
    plan = m.last_plan()
    assert plan[0] == "conv_h2" and plan.count("fc_h2_256x128") == 1 and "fc_h2_256x128_out2" in plan and "fc3_tail" in plan, plan
    rows = np.r_[0:128, 20000:20128, 32768 - 128:32768]
    ref = orc.Oracle(sd).forward_windows(orc.zscore_windows(seq[:32768 + 149])[rows])
    tol_ok(out["logits"][rows], ref["logits"], "32768-window launch")
    _argmax_ok(out["pred"][rows], ref["logits"], ref["pred"])
    full = m.infer_sequence(seq)                              # two launches: 32768 + 5000 (the fused tile's regime)
    assert np.array_equal(full["logits"][:32768], out["logits"])
    m.close()


@pytest.mark.parametrize("kind", ["aligned_rows", "aligned_rows_large_bias", "one_hot_rows"])
def test_h1_row_scale_bound_is_safe_where_it_is_tight(kind, orc):
    """h1's row scale comes from |h1| <= sqrt(K) max|feat| max_n ||W1_n||_2 + max|b1| (fc_gemm_h2_kernel<OUT2>).  Checkpoints that push h1 towards
    that bound: fc.0 rows that all point along the (non-negative) features -- the Cauchy-Schwarz case --, the same with a bias that dominates, and
    one-hot rows (h1 = single features: far BELOW the bound, the subnormal side).  No overflow, the contract holds."""
    from deep_contact_estimator_amd import synth
    sd = {k: v.copy() for k, v in synth.make_state_dict(1, "uniform").items()}
    rng = np.random.default_rng(31)
    w = sd["fc.0.weight"]
    if kind.startswith("aligned"):
        sd["fc.0.weight"] = (np.abs(rng.standard_normal(w.shape).astype(np.float32)) * np.float32(0.02) + np.float32(0.05)).astype(np.float32)
        if kind.endswith("large_bias"):
            sd["fc.0.bias"] = (sd["fc.0.bias"] * np.float32(1e6)).astype(np.float32)
        sd["fc.3.weight"] = (sd["fc.3.weight"] * np.float32(1e-2)).astype(np.float32)          # keeps the logits at an ordinary size
    else:
        z = np.zeros_like(w)
        z[np.arange(w.shape[0]), rng.integers(0, w.shape[1], w.shape[0])] = 1.0
        sd["fc.0.weight"] = z
    x = rng.standard_normal((N0, 150, 54), dtype=np.float32)
    x[7] *= np.float32(1e12)
    x[8] *= np.float32(1e-12)
    m = _model(sd, N0)
    out = m.predict(x)
    assert "fc23_fused_h2_128x64" in m.last_plan(), m.last_plan()
    rows = np.r_[0:160, N0 - 96:N0]
    ref = orc.Oracle(sd).forward_windows(x[rows])
    assert np.isfinite(ref["logits"]).all() and np.isfinite(out["logits"]).all()
    tol_ok(out["logits"][rows], ref["logits"], kind)
    _argmax_ok(out["pred"][rows], ref["logits"], ref["pred"])
    m.close()


@pytest.mark.experiments
@pytest.mark.parametrize("n", [3072, 4100])
def test_bf16_fc3_with_the_k_tiles_dealt_out_between_the_wave_groups(n):
    """(experiments build: measured no faster.)  The bf16-FC mode's fused fc.3 + fc.6 on fc_gemm_h2k_kernel<H2KFc3, FUSE6, BF16> against
    fc_gemm_phased.hip's fused 128 x 64 tile: the same bf16 products in another fp32 association -- fp32 rounding apart, a NaN window contained."""
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    a = contact_cnn(device=0, max_batch=8192, precision="bf16_fc"); a.load_state_dict(sd).eval()
    b = contact_cnn(device=0, max_batch=8192, precision="bf16_fc", tune={"bf16_fc3_ksplit": 1}); b.load_state_dict(sd).eval()
    x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
    x[n - 1, 3, 3] = np.nan
    ra, rb = a.predict(x), b.predict(x)
    assert "fc23_fused_bf16k_128x64" in b.last_plan() and "fc23_fused_bf16k_128x64" not in a.last_plan(), (a.last_plan(), b.last_plan())
    assert np.isnan(rb["logits"][n - 1]).all() and rb["pred"][n - 1] == 0
    scale = np.abs(ra["logits"][:n - 1]).max()
    assert np.abs(ra["logits"][:n - 1] - rb["logits"][:n - 1]).max() < 2e-5 * scale
    assert (ra["pred"] != rb["pred"]).sum() <= 1
    a.close(); b.close()

"""GPU (-m gpu): the DCE_FP32_SPLIT precision ("fp32_split") -- fc.0 of reference src/contact_cnn.py:48-49 at chip-filling
batches on the bf16 matrix pipe, every fp32 operand as three bf16 terms (csrc/fc_gemm_x3.hip).

The mode claims fp32 results, so it is held to the SAME contract as the fp32 path: |got - ref| <= 1e-5 max|ref| +
1e-4 |ref| against the oracle (fp64 accumulation of the reference's arithmetic), argmax exact outside the noise margin.
It does not claim the fp32 path's bits; what it must keep is everything that is not fc.0 at a large batch.
"""
import numpy as np
import pytest

from conftest import tol_ok

pytestmark = [pytest.mark.gpu, pytest.mark.experiments]      # round 6: DCE_FP32_SPLIT's kernels live in the experiments build (tests/test_experiments_gpu.py runs this file there)


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def pair():
    """(fp32 model, fp32_split model) on the bench checkpoint."""
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    a = contact_cnn(device=0, max_batch=8192); a.load_state_dict(sd).eval()
    b = contact_cnn(device=0, max_batch=8192, precision="fp32_split"); b.load_state_dict(sd).eval()
    yield sd, a, b
    a.close(); b.close()


def _argmax_ok(got_pred, ref_logits, ref_pred):
    srt = np.sort(ref_logits, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-3 * np.abs(ref_logits).max()
    assert np.array_equal(got_pred[safe], ref_pred[safe])
    flips = int((got_pred != ref_pred).sum())
    assert flips <= max(1, int(2e-3 * len(ref_pred))), flips
    return flips


def test_split_terms_are_exact():
    """a = a1 + a2 + a3 with bf16 terms, exactly, for normal fp32 values (the host routine that splits fc.0's weights is
    the device routine that splits the features)."""
    import ctypes as C
    from deep_contact_estimator_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(5)
    x = np.concatenate([rng.standard_normal(4000).astype(np.float32) * np.float32(10.0) ** rng.integers(-20, 20, 4000).astype(np.float32),
                        np.array([0.0, -0.0, 1.0, -1.0, 3.0e38, -3.0e38, 1.17549435e-38, 65504.0, 1 + 2.0 ** -23, 1 - 2.0 ** -24], np.float32)])
    planes = np.zeros((3, x.size), np.uint16)
    lib.dce_debug_split3(x.ctypes.data_as(C.c_void_p), C.c_size_t(x.size), planes.ctypes.data_as(C.c_void_p))
    terms = (planes.astype(np.uint32) << 16).view(np.float32).astype(np.float64)
    s = terms.sum(0)
    ok = np.abs(x) < 3.3e38
    assert np.array_equal(s[ok], x.astype(np.float64)[ok])
    # every term is a bf16 number by construction; the first is the value rounded to nearest-even
    u = x.view(np.uint32).astype(np.uint64)
    rne = (((u + 0x7fff + ((u >> 16) & 1)) >> 16) & 0xffff).astype(np.uint16)
    assert np.array_equal(planes[0][ok], rne[ok])


@pytest.mark.parametrize("n", [3072, 4096, 4100, 8192])
def test_split_mode_meets_the_fp32_contract(n, pair, orc):
    """Every row of a chip-filling batch against the oracle, with the tolerance of the fp32 path; the plan shows that fc.0
    really ran on the split kernel (and that 4100 = a ragged last tile went through it too)."""
    from deep_contact_estimator_amd import synth
    sd, a, b = pair
    seq = synth.make_sequence(n + 149, seed=2).astype(np.float32)
    w = b.zscore_windows(seq, 0, n)
    out = b.predict(w)
    assert "fc_x3_256x128" in b.last_plan() and b.last_plan()[0] in ("conv_x3", "conv_x3_permk", "conv_x3_permk_persist"), b.last_plan()
    ref = orc.Oracle(sd).forward_windows(w if isinstance(w, np.ndarray) else w.cpu().numpy())
    tol_ok(out["logits"], ref["logits"], f"fp32_split, {n} rows vs oracle")
    flips = _argmax_ok(out["pred"], ref["logits"], ref["pred"])
    assert np.array_equal(out["contacts"], orc.decimal2binary(out["pred"]))
    # against the fp32 MFMA path on the same input: both are fp32 evaluations, a few ulps of the largest logit apart
    nat = a.predict(w)
    d = np.abs(out["logits"].astype(np.float64) - nat["logits"]).max()
    scale = np.abs(ref["logits"]).max()
    e_split = np.abs(out["logits"].astype(np.float64) - ref["logits"]).max()
    e_nat = np.abs(nat["logits"].astype(np.float64) - ref["logits"]).max()
    print(f"n={n}: max|split - oracle| {e_split:.2e}, max|fp32 - oracle| {e_nat:.2e}, max|split - fp32| {d:.2e}, scale {scale:.2f}, flips {flips}")
    assert d < 2e-5 * scale
    assert e_split < 4 * max(e_nat, 1e-6 * scale)            # same error class as the fp32 fmaf chains, not a bf16-sized one


def test_split_mode_h1_against_fp64_on_the_device_features(pair):
    """fc.0 alone: ReLU(feat W1^T + b1) of the split kernel against an fp64 evaluation on the features the device itself
    produced -- the layer the mode replaces, isolated from everything around it."""
    from deep_contact_estimator_amd import synth
    sd, a, b = pair
    n = 3072
    x = np.random.default_rng(77).standard_normal((n, 150, 54), dtype=np.float32)
    t = b.forward_taps(x)
    assert "fc_x3_256x128" in b.last_plan()
    feat = t["feat"].astype(np.float64)
    w1 = np.asarray(sd["fc.0.weight"], np.float64)
    b1 = np.asarray(sd["fc.0.bias"], np.float64)
    rows = np.r_[0:40, 1500:1540, n - 40:n]                  # first / middle / last tiles
    ref = np.maximum(feat[rows] @ w1.T + b1, 0.0)
    tol_ok(t["h1"][rows], ref, "h1 of the split kernel vs fp64")
    # and the features themselves are the fp32 path's bits (the conv stack is untouched)
    assert np.array_equal(t["feat"], a.forward_taps(x)["feat"])


@pytest.mark.parametrize("n", [1, 30, 64, 127])
def test_split_mode_small_batches_are_the_fp32_path(n, pair):
    """Below 128 windows the mode runs the fp32 kernels throughout (latency-oriented segment / one-window conv kernels, GEMV,
    four-range MFMA kernel, chain kernel): the same bits as DCE_FP32."""
    sd, a, b = pair
    x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
    ra, rb = a.predict(x), b.predict(x)
    assert not any(k.startswith(("conv_x3", "fc_x3", "split3")) for k in b.last_plan()), b.last_plan()
    for k in ("logits", "pred", "contacts"):
        assert np.array_equal(ra[k], rb[k]), k


@pytest.mark.parametrize("n", [128, 700, 2049, 2800])
def test_split_mode_mid_size_batches(n, pair, orc):
    """From 128 windows up to the split-bf16 fc.0 kernel's threshold the mode's conv stack already runs on three-term bf16
    operands (conv_x3_kernel with fp32 features out; the FC layers stay on the fp32 kernels): every row against the oracle
    at the fp32 tolerance, a NaN window contained."""
    sd, a, b = pair
    x = np.random.default_rng(1000 + n).standard_normal((n, 150, 54), dtype=np.float32)
    x[3, 100, 0] = np.nan
    out = b.predict(x)
    plan = b.last_plan()
    assert plan[0] == "conv_x3_f32" and "fc_x3_256x128" not in plan, plan
    assert np.isnan(out["logits"][3]).all() and out["pred"][3] == 0
    keep = np.arange(n) != 3
    ref = orc.Oracle(sd).forward_windows(x[keep])
    tol_ok(out["logits"][keep], ref["logits"], f"fp32_split, mid-size batch {n}")
    _argmax_ok(out["pred"][keep], ref["logits"], ref["pred"])
    # the same rows inside a chip-filling batch come from the same conv kernel: the features, hence everything up to the
    # FC kernels' own association, agree to fp32 rounding with what the large-batch route returns
    big = b.predict(np.concatenate([x, np.zeros((3072 - n, 150, 54), np.float32)]))
    assert np.abs(big["logits"][:n][keep] - out["logits"][keep]).max() < 2e-5 * np.abs(ref["logits"]).max()


def test_split_mode_nan_window_and_sequence_path(pair, orc):
    """A window with a NaN sample yields NaN logits / class 0 for that window only (torch's behaviour, pinned for the fp32
    path by edge.npz); and the streaming entry (dce_infer_sequence, chunks of max_batch) runs the same kernels."""
    from deep_contact_estimator_amd import synth
    sd, a, b = pair
    n = 4096
    x = np.random.default_rng(3).standard_normal((n, 150, 54), dtype=np.float32)
    x[100, 7, 3] = np.nan
    out = b.predict(x)
    assert np.isnan(out["logits"][100]).all() and out["pred"][100] == 0
    keep = np.ones(n, bool); keep[100] = False
    assert np.isfinite(out["logits"][keep]).all()
    seq = synth.make_sequence(8192 + 4096 + 149, seed=11).astype(np.float32)     # chunks of 8192 and 4096 windows
    s = b.infer_sequence(seq)
    assert "fc_x3_256x128" in b.last_plan()
    ref = a.infer_sequence(seq)
    assert np.abs(s["logits"] - ref["logits"]).max() < 2e-5 * np.abs(ref["logits"]).max()
    assert (s["pred"] != ref["pred"]).sum() <= 2


@pytest.mark.parametrize("n", [3072, 4099])
def test_split_in_the_conv_kernel_equals_the_split_kernel(n, pair, monkeypatch):
    """By default the two-window conv kernel writes the features straight as three bf16 planes (plan conv_wino2_feat3);
    the option x3_unfused=1 keeps fp32 features and splits them with split3_kernel.  Same terms, same layout:
    the same bits downstream.  (4099: an odd number of rows -- the planes' stride is rounded up to even.)"""
    from deep_contact_estimator_amd import contact_cnn
    sd, a, _ = pair
    b = contact_cnn(device=0, max_batch=8192, precision="fp32_split", tune={"x3_conv": 0}); b.load_state_dict(sd).eval()     # both contexts: the fp32 Winograd conv stack
    u = contact_cnn(device=0, max_batch=8192, precision="fp32_split", tune={"x3_conv": 0, "x3_unfused": 1}); u.load_state_dict(sd).eval()
    x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
    x[5, 3, 2] = np.inf                                     # a non-finite window goes through both routes as NaN features
    rb, ru = b.predict(x), u.predict(x)
    assert "conv_wino2_feat3" in b.last_plan() and "split3" not in b.last_plan()
    assert "split3" in u.last_plan() and "conv_wino2_feat3" not in u.last_plan()
    u.close(); b.close()
    for k in ("logits", "pred", "contacts"):
        assert np.array_equal(rb[k], ru[k], equal_nan=True) if k == "logits" else np.array_equal(rb[k], ru[k]), k
    assert np.isnan(rb["logits"][5]).all()


@pytest.mark.parametrize("n,zs", [(3072, False), (4099, False), (4096, True)])
def test_conv_stack_on_three_term_operands(n, zs, pair, orc, monkeypatch):
    """The mode's conv stack too runs on the bf16 matrix pipe with three-term operands (csrc/conv_x3.hip: direct-form
    implicit GEMM on v_mfma_f32_16x16x32_bf16, one window per workgroup, activations re-split at every write-back).  Same
    contract: every row against the oracle at the fp32 tolerance -- pre-normalised windows and the fused z-score of the
    sequence entry -- and a non-finite window stays one all-NaN row."""
    from deep_contact_estimator_amd import contact_cnn, synth
    sd, a, b = pair
    m = b                                                    # conv_x3 is the mode's default conv stack
    seq = synth.make_sequence(n + 149, seed=21 + n, kind="ar1" if zs else "normal").astype(np.float32)
    if zs:
        out = m.infer_sequence(seq)
        ref = orc.Oracle(sd).infer_sequence(seq)
    else:
        w = m.zscore_windows(seq, 0, n)
        w = w if isinstance(w, np.ndarray) else w.cpu().numpy()
        w[7, 11, 5] = np.nan
        out = m.predict(w)
        ref = orc.Oracle(sd).forward_windows(np.delete(w, 7, axis=0))
        assert np.isnan(out["logits"][7]).all() and out["pred"][7] == 0
        out = {k: np.delete(v, 7, axis=0) for k, v in out.items()}
    assert m.last_plan()[0] in ("conv_x3", "conv_x3_permk") and "fc_x3_256x128" in m.last_plan(), m.last_plan()
    tol_ok(out["logits"], ref["logits"], f"conv_x3, {n} rows vs oracle")
    flips = _argmax_ok(out["pred"], ref["logits"], ref["pred"])
    err = np.abs(out["logits"].astype(np.float64) - ref["logits"])
    bound = 1e-5 * np.abs(ref["logits"]).max() + 1e-4 * np.abs(ref["logits"])
    print(f"conv_x3 n={n} zs={zs}: max err/bound {(err / bound).max():.3f}, max |err| {err.max():.2e}, flips {flips}")


@pytest.mark.parametrize("name", ["seq_normal", "seq_ar1"])
def test_conv_x3_layer_taps_vs_reference_hooks(name, golden, case_inputs, orc):
    """The layers INSIDE conv_x3_kernel -- conv1, conv2, pool1, conv3, conv4, pool2 -- against the reference's forward hooks
    (window 0 of the fixture; tests/golden/make_golden.py:96-106, reference src/contact_cnn.py:10-26,28-44) and the oracle's
    taps for two more windows, as tests/test_round3_gpu.py does for the seven fp32 conv kernel families."""
    from deep_contact_estimator_amd import contact_cnn, synth
    g = golden(name)
    sd, _ = case_inputs(g)
    m = contact_cnn(device=0, max_batch=64, precision="fp32_split"); m.load_state_dict(sd).eval()
    x = np.concatenate([g["zwin"], g["zwin"][::-1]])[:3]
    taps = m.conv_layer_taps(x, "x3")
    layers = ("conv1", "conv2", "pool1", "conv3", "conv4")
    for k in layers:
        assert not np.isnan(taps[k]).any(), (k, "positions the kernel never wrote")
        tol_ok(taps[k][0], g["tap_" + k], f"conv_x3: {k} vs the reference's forward hook")
    tol_ok(taps["feat"][0], g["tap_pool2"].reshape(-1), "conv_x3: pool2")
    o = orc.Oracle(sd)
    for i in (1, 2):
        ref = o.layer_taps(x[i])
        for k in layers:
            tol_ok(taps[k][i], ref[k], f"conv_x3: window {i} {k} vs oracle")
        tol_ok(taps["feat"][i], ref["pool2"].reshape(-1), f"conv_x3: window {i} pool2")
    # the fp32 contexts refuse the kernel (its weights exist in the split precision only)
    f = contact_cnn(device=0, max_batch=64); f.load_state_dict(sd).eval()
    with pytest.raises(RuntimeError):
        f.conv_layer_taps(x, "x3")
    f.close(); m.close()


def test_split_mode_is_deterministic_run_to_run(pair):
    """The mode's kernels order LDS-DMA, in-place LDS write-backs and fragment reads with counted waits and barriers only: a
    race would show as run-to-run differences.  40 repeats of a ragged batch, alternating two contexts (tools/race_screen_x3.py
    runs 755 over five sizes with a competing copy stream: profiles/r3d_race_screen_x3.txt)."""
    from deep_contact_estimator_amd import contact_cnn
    sd, a, b = pair
    c = contact_cnn(device=0, max_batch=8192, precision="fp32_split"); c.load_state_dict(sd).eval()
    x = np.random.default_rng(41).standard_normal((4100, 150, 54), dtype=np.float32)
    ref = b.predict(x)["logits"].copy()
    for r in range(40):
        got = (c if r % 2 else b).predict(x)["logits"]
        assert np.array_equal(got, ref), r
    c.close()

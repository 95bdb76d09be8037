"""GPU (-m gpu), round 4: the chip-filling kernels against the REFERENCE's own numbers (tests/golden/chip_ar1.npz: 4096 AR(1)
windows through the imported reference, make_golden.py case C; reference src/test.py:72-107), the three-term conv stack for
chip-filling batches (csrc/conv_x3p.hip) against conv_x3.hip and the oracle, batch-size regimes of the non-default precisions."""
import os

import numpy as np
import pytest

from conftest import needs_experiments, tol_ok

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _model(precision, max_batch=8192, tune=None):
    from deep_contact_estimator_amd import contact_cnn
    return contact_cnn(device=0, max_batch=max_batch, precision=precision, tune=tune)


@pytest.mark.parametrize("precision", ["fp32", pytest.param("fp32_split", marks=needs_experiments), "fp32_f16x2", "bf16_fc"])
def test_chip_filling_launch_vs_the_reference(precision, golden, case_inputs, orc):
    """One 4096-window launch per precision against logits the reference itself produced: fp32, fp32_split and fp32_f16x2 within the
    fp32 tolerance and argmax-exact outside the noise margin, bf16_fc within its band (bf16 operands of fc.0 / fc.3)."""
    g = golden("chip_ar1")
    sd, seq = case_inputs(g)
    m = _model(precision, max_batch=4096)
    m.load_state_dict(sd).eval()
    out = m.infer_sequence(seq)
    plan = m.last_plan()
    ref, scale = g["logits"], np.abs(g["logits"]).max()
    safe = g["margin"] > 1e-3 * scale
    if precision == "bf16_fc":
        err = np.abs(out["logits"] - ref).max()
        flips = out["pred"] != g["pred"]
        assert err < 2e-2 * scale, (err, scale)
        assert flips.mean() < 0.02 and (g["margin"][flips] < 4 * err + 1e-6).all()
        assert plan[0] == "conv_h2_bf16_permk" and "fc_phased256x128_bf16" in plan, plan         # conv results of fp32 grade (two fp16 terms, per-window scales), bf16 FC
    else:
        # Against the fp64-statistics evaluation (the oracle) on the same rows: the contract as stated, every logit.  Against the
        # reference's own numbers: the reference z-scores in fp32 and on this sequence sits up to 1.15 bounds from that evaluation
        # itself (five named logits outside the contract, tests/conftest.py CHIP_AR1_REFERENCE_ZSCORE_OUTLIERS), so a logit may be
        # as far from the reference as the contract plus the reference's own distance from the fp64 evaluation -- no blanket factor.
        o = orc.Oracle(sd).infer_sequence(seq)
        tol_ok(out["logits"], o["logits"], f"{precision}: 4096 AR(1) windows vs the fp64-statistics evaluation")
        r64 = ref.astype(np.float64)
        bound = 1e-5 * np.abs(r64).max() + 1e-4 * np.abs(r64)
        over = np.abs(out["logits"].astype(np.float64) - r64) - (bound + np.abs(o["logits"].astype(np.float64) - r64))
        assert (over <= 0).all(), f"{precision}: {(over > 0).sum()} logits further from the reference than the contract + the reference's own z-score noise"
        assert np.array_equal(out["pred"][safe], g["pred"][safe])
        assert (out["pred"] != g["pred"]).sum() <= 2
        if precision == "fp32": assert "fc_phased256x128" in plan and "fc23_fused_phased128x64" in plan, plan
        elif precision == "fp32_f16x2": assert plan[:3] == ["conv_h2", "fc_h2_256x128_out2", "fc23_fused_h2_128x64"], plan
        else: assert "fc_x3_256x128" in plan and any(k.startswith("conv_x3") for k in plan), plan
    assert np.array_equal(out["contacts"], ((out["pred"][:, None] & np.array([8, 4, 2, 1])) != 0).astype(np.uint8))
    m.close()


@pytest.mark.experiments                                        # (the three-term form it is compared with lives in the experiments build since round 6)
@pytest.mark.parametrize("n", [128, 515, 4096, 4099])
def test_bf16_fc_conv_stack_on_two_term_operands(n, orc):
    """DCE_BF16_FC's conv stack up to 256 windows per launch, in the online pushes and with the option bf16_conv_h2=0: conv_x3.hip with NT = 2 -- operands as two bf16 terms, a1 b1 + a1 b2 + a2 b1 (three MFMAs per
    product, ~17 significant bits), two LDS planes, three workgroups per CU -- against the same kernel on three-term operands
    (option x3_bf16_terms=3, fp32-grade).  The features leave rounded to bf16 (8 bits), so the two may differ only where a value sits
    within ~2^-17 of a rounding boundary: few values, one bf16 ulp each; the logits agree far inside the mode's band and the error
    against the fp64 oracle is the same.  Non-finite windows, odd sizes, both feature orders (a tap keeps the reference's flatten
    order), the z-score entry."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model("bf16_fc", tune={"bf16_conv_h2": 0}); a.load_state_dict(sd).eval()
    b = _model("bf16_fc", tune={"x3_bf16_terms": 3}); b.load_state_dict(sd).eval()
    x = np.random.default_rng(50 + n).standard_normal((n, 150, 54), dtype=np.float32)
    x[n // 2, 3, 7] = np.inf
    ta, tb = a.forward_taps(x), b.forward_taps(x)
    assert a.last_plan()[0] == "conv_x2_bf16" and b.last_plan()[0] == "conv_x3_bf16", (a.last_plan(), b.last_plan())
    fa, fb = orc.bf16_from_bits(ta["feat"]), orc.bf16_from_bits(tb["feat"])
    assert np.isnan(fa[n // 2]).all() and np.isnan(fb[n // 2]).all()
    fa, fb = np.delete(fa, n // 2, 0), np.delete(fb, n // 2, 0)
    d = fa != fb
    assert d.mean() < 2e-2, d.mean()
    _, e = np.frexp(np.maximum(np.abs(fa), np.abs(fb)))
    # one bf16 ulp (8 significand bits); next to zero (ReLU'd sums that cancel) the two-term products' absolute error, ~2^-17 of the
    # sum of the products' magnitudes, is what separates the two
    assert (np.abs(fa - fb)[d] <= np.ldexp(1.0, e[d] - 8) + 1e-4 * np.abs(fb).max()).all()
    ra, rb = a.predict(x), b.predict(x)
    assert a.last_plan()[0] == "conv_x2_bf16_permk" and b.last_plan()[0] == "conv_x3_bf16_permk"
    ok = np.arange(n) != n // 2
    assert np.isnan(ra["logits"][n // 2]).all() and ra["pred"][n // 2] == 0
    ref = orc.Oracle(sd).forward_windows(x[ok][:600])
    scale = np.abs(ref["logits"]).max()
    ea, eb = (np.abs(r["logits"][ok][:600].astype(np.float64) - ref["logits"]).max() for r in (ra, rb))
    assert ea < 2e-2 * scale and ea < 1.5 * eb + 1e-3 * scale, (ea, eb, scale)
    assert np.abs(ra["logits"][ok].astype(np.float64) - rb["logits"][ok]).max() < 5e-3 * scale
    seq = synth.make_sequence(n + 149, seed=n).astype(np.float32)
    sa, sb = a.infer_sequence(seq), b.infer_sequence(seq)
    assert a.last_plan()[0] == "conv_x2_bf16_permk"
    assert np.abs(sa["logits"].astype(np.float64) - sb["logits"]).max() < 5e-3 * np.abs(sb["logits"]).max()
    a.close(); b.close()


@pytest.mark.experiments
def test_k_tiles_dealt_out_between_the_wave_groups():
    """fc_gemm_ki_kernel (option gemm_ki=1, experiments build: group 0 multiplies the even K-tiles, group 1 the odd ones, partial sums meet
    once through LDS; same launch time as the phased kernel at a lower clock, profiles/r4o_gemm_ki.txt): another fp32 summation order,
    so h1 may differ at bf16 rounding boundaries -- the logits stay within the mode's batch-size band of the phased kernel's, and every
    repeat gives the same bytes."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model("bf16_fc", tune={"gemm_ki": 1}); a.load_state_dict(sd).eval()
    b = _model("bf16_fc", tune={"gemm_ki": 0}); b.load_state_dict(sd).eval()
    for n in (4096, 8192):
        x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
        ra, rb = a.predict(x), b.predict(x)
        assert "fc_ki256x128" in a.last_plan() and "fc_phased256x128_bf16" in b.last_plan(), (a.last_plan(), b.last_plan())
        assert np.abs(ra["logits"] - rb["logits"]).max() <= 2e-3 * np.abs(rb["logits"]).max()
        for _ in range(10):
            assert np.array_equal(a.predict(x)["logits"], ra["logits"])
    a.close(); b.close()


@pytest.mark.experiments
def test_barrier_free_bf16_gemm_equals_the_phased_kernel():
    """fc_gemm_pipe_kernel (option gemm_pipe=1, experiments build: LDS counters instead of workgroup barriers in the K loop; measured
    slower, profiles/r4j_gemm_pipe.txt) walks K in the same 16-k blocks as the phased kernel: the same bytes, on every repeat."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model("bf16_fc", tune={"gemm_pipe": 1}); a.load_state_dict(sd).eval()
    b = _model("bf16_fc", tune={"gemm_pipe": 0}); b.load_state_dict(sd).eval()
    for n in (4096, 8192):
        x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
        ra, rb = a.predict(x), b.predict(x)
        assert "fc_pipe256x128" in a.last_plan() and "fc_phased256x128_bf16" in b.last_plan(), (a.last_plan(), b.last_plan())
        assert np.array_equal(ra["logits"], rb["logits"])
        for _ in range(20):
            assert np.array_equal(a.predict(x)["logits"], ra["logits"])
    a.close(); b.close()


@pytest.mark.experiments
@pytest.mark.parametrize("precision", [pytest.param("fp32_split", marks=needs_experiments), "bf16_fc"])
def test_persistent_conv_stack_equals_one_workgroup_per_window(precision):
    """conv_x3.hip's persistent form (option x3_persist=1, experiments build: two workgroups per CU walk the windows, the next window's
    samples requested a layer ahead) runs the same arithmetic in the same order: the same BYTES as one workgroup per window, at
    ragged sizes (workgroups with different window counts, fewer windows than workgroups), on both entries, with bad windows."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model(precision, tune={"x3_persist": 1, "x3_persist_min": 128, "x3_bf16_terms": 3}); a.load_state_dict(sd).eval()
    b = _model(precision, tune={"x3_persist": 0, "x3_bf16_terms": 3}); b.load_state_dict(sd).eval()
    for n in (300, 4096, 4097, 5001):
        if precision == "fp32_split" and n < 2817: continue
        x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
        x[n // 3, 17, 3] = np.nan; x[n - 1, 149, 53] = np.inf
        ra, rb = a.predict(x), b.predict(x)
        assert a.last_plan()[0].endswith("_persist") and not b.last_plan()[0].endswith("_persist"), (a.last_plan(), b.last_plan())
        assert np.array_equal(ra["logits"], rb["logits"], equal_nan=True) and np.array_equal(ra["pred"], rb["pred"]), n
        assert np.isnan(ra["logits"][n // 3]).all() and np.isnan(ra["logits"][n - 1]).all() and np.isfinite(ra["logits"][0]).all()
    seq = synth.make_sequence(6000 + 149, seed=3).astype(np.float32)
    sa, sb = a.infer_sequence(seq), b.infer_sequence(seq)
    assert a.last_plan()[0].endswith("_persist")
    assert np.array_equal(sa["logits"], sb["logits"])
    a.close(); b.close()


@pytest.mark.experiments
@pytest.mark.parametrize("precision", [pytest.param("fp32_split", marks=needs_experiments), "bf16_fc"])
def test_paired_conv_stack_vs_one_window_kernel_and_oracle(precision, orc):
    """conv_x3p.hip (option x3_pair=1: one 8-wave workgroup per CU, two windows per wave, write-backs inside the other window's K
    loops, features in the K order t' * 128 + c with fc.0's weights permuted alike) against conv_x3.hip on the same windows:
    ragged and odd batch sizes (a pair's second window missing, workgroups with different pair counts), the z-score entry,
    non-finite windows; fp32_split also against the oracle at the fp32 tolerance."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model(precision, tune={"x3_pair": 1, "x3_pair_min": 256}); a.load_state_dict(sd).eval()
    b = _model(precision, tune={"x3_pair": 0}); b.load_state_dict(sd).eval()
    lim = 2e-5 if precision == "fp32_split" else 2e-2
    for n in (256, 511, 4096, 4097, 5001):
        if precision == "fp32_split" and n < 2817: continue                # (below the split fc.0's threshold the mode keeps fp32 features)
        x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
        ra, rb = a.predict(x), b.predict(x)
        assert a.last_plan()[0].startswith("conv_x3p") and not b.last_plan()[0].startswith("conv_x3p"), (a.last_plan(), b.last_plan())
        scale = np.abs(rb["logits"]).max()
        assert np.abs(ra["logits"].astype(np.float64) - rb["logits"]).max() < lim * scale, n
        if precision == "fp32_split" and n == 4097:
            ref = orc.Oracle(sd).forward_windows(x)
            tol_ok(ra["logits"], ref["logits"], "conv_x3p + fc_x3 vs oracle")
            srt = np.sort(ref["logits"], axis=1)
            safe = (srt[:, -1] - srt[:, -2]) > 1e-3 * np.abs(ref["logits"]).max()
            assert np.array_equal(ra["pred"][safe], ref["pred"][safe])
    seq = synth.make_sequence(6000 + 149, seed=3).astype(np.float32)
    sa, sb = a.infer_sequence(seq), b.infer_sequence(seq)
    assert a.last_plan()[0].startswith("conv_x3p")
    assert np.abs(sa["logits"].astype(np.float64) - sb["logits"]).max() < lim * np.abs(sb["logits"]).max()
    x = np.random.default_rng(9).standard_normal((4096, 150, 54), dtype=np.float32)
    x[5, 17, 3] = np.nan; x[1000, 149, 53] = np.inf; x[4095, 0, 0] = -np.inf
    r = a.predict(x)
    bad = np.isnan(r["logits"]).all(1)
    assert bad[[5, 1000, 4095]].all() and bad.sum() == 3 and (r["pred"][bad] == 0).all()
    again = a.predict(x)
    assert np.array_equal(r["logits"], again["logits"], equal_nan=True)             # run-to-run determinism
    a.close(); b.close()


@pytest.mark.parametrize("precision", [pytest.param("fp32_split", marks=needs_experiments), "bf16_fc"])
def test_batch_size_regimes_stay_within_the_mode_tolerance(precision, orc):
    """The non-default precisions pick their conv / fc.0 kernels by the size of the launch (below 128 windows the fp32 kernels,
    from 128 the three-term conv stack, from 2817 the split fc.0): the same windows in a small and in a large launch may differ
    in the last bits, never by more than the mode's tolerance (include/dce.h, "batch-size regimes")."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    m = _model(precision); m.load_state_dict(sd).eval()
    x = np.random.default_rng(4).standard_normal((4096, 150, 54), dtype=np.float32)
    big = m.predict(x)
    plan_big = m.last_plan()
    ref = orc.Oracle(sd).forward_windows(x[:256])
    for n in (64, 127, 128, 256):
        small = m.predict(x[:n])
        assert m.last_plan() != plan_big or n >= 2817
        d = np.abs(small["logits"].astype(np.float64) - big["logits"][:n]).max()
        scale = np.abs(ref["logits"]).max()
        if precision == "fp32_split":
            tol_ok(small["logits"], ref["logits"][:n], f"{n} windows vs oracle")
            assert d < 2e-5 * scale
        else:
            assert d < 2e-2 * scale
    m.close()


def test_packed_rows_must_be_word_aligned():
    """The kernels write and read packed rows as 32-bit words: a device buffer at an odd byte offset is refused with DCE_ERR_ARG by
    every entry point that takes one (include/dce.h), not handed to a kernel."""
    import torch
    from deep_contact_estimator_amd import contact_cnn, synth
    m = contact_cnn(device=0, max_batch=64); m.load_state_dict(synth.make_state_dict(1)).eval()
    x = torch.randn((8, 150, 54), device="cuda")
    buf = torch.empty(8 * 68 + 8, dtype=torch.uint8, device="cuda")
    odd = buf[1:1 + 8 * 68].view(8, 68)
    assert odd.data_ptr() % 4 == 1 and odd.is_contiguous()
    with pytest.raises(RuntimeError, match="4-byte aligned"):
        m.predict_packed(x, out=odd)
    good = m.predict_packed(x)
    odd.copy_(good)
    with pytest.raises(RuntimeError, match="4-byte aligned"):
        m.unpack_results(odd)
    m.close()


def test_layer_taps_refuse_a_bf16_fc_context():
    from deep_contact_estimator_amd import contact_cnn, synth
    m = contact_cnn(device=0, max_batch=64, precision="bf16_fc"); m.load_state_dict(synth.make_state_dict(1)).eval()
    x = np.random.default_rng(0).standard_normal((2, 150, 54), dtype=np.float32)
    with pytest.raises(RuntimeError, match="DCE_BF16_FC"):
        m.conv_layer_taps(x)
    m.close()


@pytest.mark.parametrize("precision", [pytest.param("fp32_split", marks=needs_experiments), "bf16_fc"])
def test_features_straight_from_the_accumulators_equal_the_staged_ones(precision):
    """conv_x3.hip writes its features straight from the accumulators in the K order t' * 128 + c (option x3_permk, default) with fc.0's
    weights permuted alike, or through LDS in the reference's flatten order (x3_permk=0): the same products, another order of the
    fc.0 summation -- logits agree within the mode's noise, NaN windows are contained either way, a feature tap still hands out
    the reference's order."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    a = _model(precision, tune={"x3_permk": 1}); a.load_state_dict(sd).eval()
    b = _model(precision, tune={"x3_permk": 0}); b.load_state_dict(sd).eval()
    for n in (4096, 4099, 3000 if precision == "fp32_split" else 129):
        x = np.random.default_rng(n).standard_normal((n, 150, 54), dtype=np.float32)
        x[3, 0, 0] = np.nan
        ra, rb = a.predict(x), b.predict(x)
        assert a.last_plan()[0].endswith("_permk") and not b.last_plan()[0].endswith("_permk"), (a.last_plan(), b.last_plan())
        scale = np.nanmax(np.abs(rb["logits"]))
        d = np.nanmax(np.abs(ra["logits"].astype(np.float64) - rb["logits"]))
        assert d < (2e-5 if precision == "fp32_split" else 2e-2) * scale, (n, d)
        assert np.isnan(ra["logits"][3]).all() and np.isnan(rb["logits"][3]).all() and np.isfinite(ra["logits"][4]).all()
    t = a.forward_taps(np.random.default_rng(1).standard_normal((256, 150, 54), dtype=np.float32))
    assert not a.last_plan()[0].endswith("_permk")                       # a feature tap keeps the reference's flatten order
    a.close(); b.close()

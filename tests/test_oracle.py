"""CPU: pin the oracle (C restatement + torch functional restatement) against the golden
vectors captured from the imported reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import tol_ok
from oracle import oracle as orc

CASES = ["seq_normal", "seq_ar1"]


@pytest.mark.parametrize("name", CASES)
def test_zscore_matches_contact_dataset(name, golden, case_inputs):
    g = golden(name)
    _, seq = case_inputs(g)
    w = orc.zscore_windows(seq)
    assert w.shape == (seq.shape[0] - 149, 150, 54)
    # reference: utils/data_handler.py:55-56 on fp32 data; fp32 round-off of mean/std only
    np.testing.assert_allclose(w[g["zwin_idx"]], g["zwin"], rtol=0, atol=2e-5 * np.abs(g["zwin"]).max())


@pytest.mark.parametrize("name", CASES)
def test_sequence_logits_argmax_contacts(name, golden, case_inputs):
    g = golden(name)
    sd, seq = case_inputs(g)
    out = orc.Oracle(sd).infer_sequence(seq)
    tol_ok(out["logits"], g["logits"], "logits")
    # argmax must be exact wherever the reference's own top-2 margin clears the fp32 noise floor
    tau = 1e-3 * np.abs(g["logits"]).max()
    safe = g["margin"] > tau
    assert safe.sum() >= 0.95 * safe.size
    assert np.array_equal(out["pred"][safe], g["pred"][safe])
    assert np.array_equal(out["contacts"][safe], g["contacts"][safe])
    # in these fixtures nothing sits below the floor either: whole arrays bit-identical
    assert np.array_equal(out["pred"], g["pred"])
    assert np.array_equal(out["contacts"], g["contacts"])


@pytest.mark.parametrize("name", CASES)
def test_layer_taps_window0(name, golden, case_inputs):
    g = golden(name)
    sd, seq = case_inputs(g)
    o = orc.Oracle(sd)
    win0 = g["zwin"][0]                       # the reference's own z-scored window 0
    taps = o.layer_taps(win0)
    for k in ("conv1", "conv2", "pool1", "conv3", "conv4", "pool2"):
        tol_ok(taps[k], g["tap_" + k], k)
    fw = o.forward_windows(win0[None], taps=True)
    # flatten is channel-major c*37+t (src/contact_cnn.py:64)
    tol_ok(fw["feat"][0], g["tap_pool2"].reshape(-1), "feat")
    tol_ok(fw["h1"][0], g["tap_fc1"], "fc1")
    tol_ok(fw["h2"][0], g["tap_fc2"], "fc2")
    tol_ok(fw["logits"][0], g["logits"][0], "logits0")


def test_edge_semantics(golden):
    from deep_contact_estimator_amd import synth
    e = golden("edge")
    # decimal2binary table, MSB = leg 0 (src/inference_one_seq.py:59-62)
    assert np.array_equal(orc.decimal2binary(np.arange(16)), e["decimal2binary"])
    assert np.array_equal(orc.decimal2binary(np.array([9]))[0], [1, 0, 0, 1])
    # ties -> lowest index
    assert np.array_equal(orc.argmax16(e["tie_logits"]), e["tie_pred"])
    # constant channel -> NaN column -> NaN logits -> class 0
    seq = synth.make_sequence(int(e["const_T"]), int(e["const_sseed"]), "normal")
    seq[:, int(e["const_channel"])] = float(e["const_value"])
    sd = synth.make_state_dict(1, "uniform")
    out = orc.Oracle(sd).infer_sequence(seq.astype(np.float32), want_windows=True)
    assert np.array_equal(np.isnan(out["windows"]).all(axis=(0, 1)), e["const_zwin_nan_cols"])
    assert np.array_equal(np.isnan(out["logits"]), e["const_logits_isnan"])
    assert np.array_equal(out["pred"], e["const_pred"])


def test_empty_and_short_sequences():
    from deep_contact_estimator_amd import synth
    o = orc.Oracle(synth.make_state_dict(1))
    for T in (0, 1, 149):
        out = o.infer_sequence(np.zeros((T, 54), np.float32))
        assert out["logits"].shape == (0, 16) and out["contacts"].shape == (0, 4)
    out = o.infer_sequence(synth.make_sequence(150, 3).astype(np.float32))
    assert out["logits"].shape == (1, 16)


@pytest.mark.parametrize("name", CASES)
def test_torch_restatement_matches_reference(name, golden, case_inputs):
    """oracle/torch_ref.py is what bench.py times as cpu_baseline; it must BE the reference path."""
    torch = pytest.importorskip("torch")
    from oracle import torch_ref
    g = golden(name)
    sd, seq = case_inputs(g)
    torch.set_num_threads(1)
    tsd = torch_ref.to_torch(sd)
    contacts = torch_ref.reference_loop(tsd, torch.from_numpy(seq), int(g["batch"])).numpy()
    assert np.array_equal(contacts, g["contacts"])
    logits = torch_ref.forward(tsd, torch.from_numpy(orc.zscore_windows(seq))).numpy()
    tol_ok(logits, g["logits"], "torch_ref logits")


def test_oracle_vs_reference_loop_functions(golden, case_inputs):
    """The contacts the reference's own inference() / inference_and_compute_acc() return
    (src/inference_one_seq.py:19-30,33-57; fixture loop_one_seq.npz) and its accuracy numbers at
    the shipped batch_size 1, from the oracle's argmax + the elementwise definition."""
    from oracle import oracle
    g = golden("loop_one_seq")
    sd, seq = case_inputs(g)
    out = oracle.Oracle(sd).infer_sequence(seq)
    assert np.array_equal(out["contacts"], g["contacts_B1"]) and np.array_equal(out["contacts"], g["contacts_B30"])
    gt = g["labels"].reshape(-1)[149:]
    assert (out["pred"] == gt).mean() == float(g["acc_B1"])
    assert np.array_equal((out["contacts"] == oracle.decimal2binary(gt)).mean(0), g["acc_per_leg_B1"])
    assert float(g["acc_B30"]) > 1.0          # (B,)==(B,1) broadcast at :54 -- not an accuracy; kept as documentation


def test_bf16fc_restatement_pinned_on_reference_derived_vectors(golden, case_inputs):
    """oracle_forward_windows_bf16fc (BASELINE configs[4]) against tests/golden/bf16fc.npz, which make_golden.py derived
    from the reference's own modules with torch's bfloat16 conversion and float64 matmuls.  Stage by stage on the
    golden's own layer inputs the two fp64-accumulated evaluations must agree to the last bf16 / fp32 bit pattern
    (bf16 products are exact in fp64); end to end (features from the oracle's conv stack instead of MKL-DNN's) a few
    features round the other way at bf16 boundaries, hence the band."""
    g = golden("bf16fc")
    sd, seq = case_inputs(g)
    w1, w2 = orc.bf16_round(sd["fc.0.weight"]), orc.bf16_round(sd["fc.3.weight"])
    # torch's own conversion of the weights agrees with the restated rounding, bit for bit
    import torch
    for k, w in (("fc.0.weight", w1), ("fc.3.weight", w2)):
        t = torch.from_numpy(sd[k]).to(torch.bfloat16).float().numpy()
        assert np.array_equal(t.view(np.uint32), w.view(np.uint32)), k
    feat = orc.bf16_from_bits(g["feat_bf16_w0"])[None]
    h1 = orc.linear_rows(feat, w1, sd["fc.0.bias"], relu=True)
    assert np.array_equal(orc.bf16_bits(orc.bf16_round(h1))[0], g["h1_bf16_w0"])
    h2 = orc.linear_rows(orc.bf16_round(h1), w2, sd["fc.3.bias"], relu=True)
    np.testing.assert_allclose(h2[0], g["h2_w0"], rtol=1e-6, atol=1e-6 * np.abs(g["h2_w0"]).max())
    lg = orc.linear_rows(h2, sd["fc.6.weight"], sd["fc.6.bias"], relu=False)
    np.testing.assert_allclose(lg[0], g["logits"][0], rtol=1e-6, atol=1e-6 * np.abs(g["logits"]).max())
    # end to end from the raw sequence
    o = orc.Oracle(sd, bf16_fc=True)
    out = o.forward_windows(orc.zscore_windows(seq), taps=True)
    scale = np.abs(g["logits"]).max()
    assert np.abs(out["logits"] - g["logits"]).max() <= 1e-3 * scale
    fb = orc.bf16_bits(out["feat"][0])
    assert (fb != g["feat_bf16_w0"]).mean() < 5e-3                     # rounding-boundary flips only
    assert np.array_equal(orc.bf16_round(out["feat"]), out["feat"]) and np.array_equal(orc.bf16_round(out["h1"]), out["h1"])
    srt = np.sort(g["logits"], axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-2 * scale
    assert safe.sum() > 0.8 * safe.size and np.array_equal(out["pred"][safe], g["pred"][safe])
    # and the mode is a real change of arithmetic: it differs from fp32 by ~bf16 epsilon, not by fp32 noise
    d = np.abs(g["logits"] - g["fp32_logits"]).max()
    assert 1e-4 * scale < d < 5e-2 * scale


def test_bf16_round_edge_cases():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.01171875, -1.00390625, 3.4e38, np.inf, -np.inf, 0.0, -0.0, 1e-40],
                 np.float32)
    r = orc.bf16_round(x)
    # ties to even: 1 + 2^-8 is half way between 1 and 1 + 2^-7 -> 1 (even); 1 + 3*2^-9 rounds up; 1 + 3*2^-8 ties -> 1 + 2^-6
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0078125 and r[3] == 1.015625 and r[4] == -1.0
    assert np.isinf(r[5]) and np.isinf(r[6]) and r[7] == -np.inf and r[8] == 0 and np.signbit(r[9])
    assert np.isnan(orc.bf16_round(np.array([np.nan], np.float32))[0])
    import torch
    big = np.random.default_rng(0).standard_normal(200000).astype(np.float32) * np.float32(37.0)
    assert np.array_equal(orc.bf16_round(big).view(np.uint32), torch.from_numpy(big).to(torch.bfloat16).float().numpy().view(np.uint32))


def test_chip_filling_golden_pins_the_oracle(golden, case_inputs):
    """tests/golden/chip_ar1.npz: 4096 AR(1) windows through the reference's own model and dataset (make_golden.py case C,
    src/test.py:72-107 loop shape) -- the launch size of BASELINE configs[1] against the reference's numbers, not only against
    the restatement."""
    g = golden("chip_ar1")
    sd, seq = case_inputs(g)
    out = orc.Oracle(sd).infer_sequence(seq)
    assert out["logits"].shape == (4096, 16)
    # The reference z-scores in fp32 (utils/data_handler.py:55-56); on this sequence -- offsets up to 5 on spreads down to 0.01 --
    # its fp32 mean alone is off by up to 5e-5 standard deviations.  Against the fp64-statistics evaluation that leaves FIVE of its
    # 65,536 logits just outside the contract (1.001 .. 1.146 bounds; 41 windows come within 0.7 of it, the 128-window AR(1) fixture
    # stays inside): those five are named, every other logit is held to the contract as stated.
    from conftest import CHIP_AR1_REFERENCE_ZSCORE_OUTLIERS as named
    ref = g["logits"].astype(np.float64)
    ratio = np.abs(out["logits"].astype(np.float64) - ref) / (1e-5 * np.abs(ref).max() + 1e-4 * np.abs(ref))
    outside = sorted((int(a), int(b)) for a, b in np.argwhere(ratio > 1.0))
    assert outside == sorted(named), outside
    assert max(ratio[a, b] for a, b in named) < 1.2
    safe = g["margin"] > 1e-3 * np.abs(g["logits"]).max()
    assert safe.sum() >= 0.95 * safe.size
    assert np.array_equal(out["pred"][safe], g["pred"][safe])
    assert (out["pred"] != g["pred"]).sum() <= 2

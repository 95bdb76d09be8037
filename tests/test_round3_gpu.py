"""GPU (-m gpu): round-3 parity additions.

* the layers INSIDE the fused conv stack, for every conv kernel family, against the reference's forward-hook
  goldens (tests/golden/make_golden.py:96-106; reference src/contact_cnn.py:10-26,28-44);
* the bf16-FC mode (BASELINE.json configs[4]) against an independent restatement, stage by stage on the device's
  own inputs of each layer and end to end (oracle/dce_oracle.c oracle_forward_windows_bf16fc);
* packed (n,68) result rows (the gather's wire format) against the three reference-shaped arrays;
* the per-context A/B switches really select different kernels (dce_last_plan).
"""
import numpy as np
import pytest

from conftest import needs_experiments, tol_ok

pytestmark = pytest.mark.gpu

CASES = ["seq_normal", "seq_ar1"]
# kernel family -> how many windows the test feeds it (the two-window families get two workgroups' worth)
CONV_KERNELS = {"wino2rt4": 4, "wino2": 4, "wino1x8": 3, "half": 3, "quarter": 3, "direct": 4, "wino1x4": 2}
LAYERS = ("conv1", "conv2", "pool1", "conv3", "conv4")


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def model_of():
    from deep_contact_estimator_amd import contact_cnn, synth
    cache = {}

    def get(wseed=1, bias="uniform", precision="fp32", max_batch=4096, tune=None):
        key = (wseed, bias, precision, max_batch, str(tune))
        if key not in cache:
            m = contact_cnn(device=0, max_batch=max_batch, precision=precision, tune=tune)
            m.load_state_dict(synth.make_state_dict(wseed, bias))
            cache[key] = m.eval()
        return cache[key]
    yield get
    for m in cache.values():
        m.close()


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("kernel", list(CONV_KERNELS))
def test_conv_layer_taps_vs_reference_hooks(kernel, name, golden, case_inputs, model_of, orc):
    """conv1, conv2, pool1, conv3, conv4 of the reference's window 0 (forward hooks on block1[1], block1[3], block1[5],
    block2[1], block2[3]) for EVERY conv kernel family, and the oracle's taps for the other windows of the launch
    (so that both windows of a two-window workgroup and every segment of a cut window are covered)."""
    from conftest import has_experiments
    if kernel in ("wino2rt4", "direct", "wino1x4") and not has_experiments():
        pytest.skip("the four-row-tile workgroup, the four-wave one-window kernel and the direct-form kernel live in the experiments build (tests/test_experiments_gpu.py runs them there)")
    g = golden(name)
    sd, _ = case_inputs(g)
    m = model_of(int(g["wseed"]), str(g["bias"]))
    nwin = CONV_KERNELS[kernel]
    x = np.concatenate([g["zwin"], g["zwin"][::-1]])[:nwin]          # window 0 first; 4 distinct windows, then repeats
    taps = m.conv_layer_taps(x, kernel)
    seg = kernel in ("half", "quarter")                              # these never compute conv4's t = 74 (the pool drops it)
    for k in LAYERS:
        got = taps[k][0]
        ref = g["tap_" + k]
        if k == "conv4" and seg:
            assert np.isnan(got[:, 74]).all()
            got, ref = got[:, :74], ref[:, :74]
        assert not np.isnan(got).any(), (kernel, k, "positions the kernel never wrote")
        tol_ok(got, ref, f"{kernel}: {k} vs the reference's forward hook")
    tol_ok(taps["feat"][0], g["tap_pool2"].reshape(-1), f"{kernel}: pool2")
    o = orc.Oracle(sd)
    for i in range(1, nwin):
        ref = o.layer_taps(x[i])
        for k in LAYERS:
            got, want = taps[k][i], ref[k]
            if k == "conv4" and seg:
                got, want = got[:, :74], want[:, :74]
            tol_ok(got, want, f"{kernel}: window {i} {k} vs oracle")
        tol_ok(taps["feat"][i], ref["pool2"].reshape(-1), f"{kernel}: window {i} pool2")


def test_conv_layer_taps_bit_identical_across_winograd_families(golden, model_of):
    """Every Winograd kernel family walks a layer's K in the same order per accumulator: not only the features
    (test_small_batch_kernels_are_bit_identical) but every intermediate layer is the same bits."""
    g = golden("seq_ar1")
    m = model_of(int(g["wseed"]), str(g["bias"]))
    x = g["zwin"][:3]
    base = m.conv_layer_taps(x, "wino2")
    from conftest import has_experiments
    for kernel in (("wino2rt4", "wino1x4") if has_experiments() else ()) + ("wino1x8", "half", "quarter"):
        t = m.conv_layer_taps(x, kernel)
        for k in LAYERS + ("feat",):
            a, b = base[k], t[k]
            if k == "conv4" and kernel in ("half", "quarter"):
                a, b = a[:, :, :74], b[:, :, :74]
            assert np.array_equal(a, b), (kernel, k)


def test_tapped_kernels_produce_the_product_features(model_of):
    """The TAPS instantiations differ from the shipped kernels only by extra stores: same features, bit for bit."""
    m = model_of()
    x = np.random.default_rng(3).standard_normal((5, 150, 54), dtype=np.float32)
    feat = m.forward_taps(x)["feat"]
    from conftest import has_experiments
    for kernel in (("wino2rt4", "wino1x4") if has_experiments() else ()) + ("wino2", "wino1x8", "half", "quarter"):
        assert np.array_equal(m.conv_layer_taps(x, kernel)["feat"], feat), kernel


# ------------------------------------------------------------------------------------------------
# bf16-FC mode: independent restatement
# ------------------------------------------------------------------------------------------------
# batch sizes chosen per bf16 kernel: phased 256x128 + fused 128x64 (4096), phased 128x64 (3000 / 700), 64x64 tiles (300), the
# weight-streaming kernel with one, two and four 64-window blocks (1 / 40, 100, 256), a ragged size past a round (4100)
@pytest.mark.parametrize("terms", ["h2", 2, pytest.param(3, marks=needs_experiments)])                     # the mode as it ships (conv results of fp32 grade from two fp16 terms with per-window scales above 256 windows: BASELINE configs[4] as written) / two bf16 terms at every size (bf16_conv_h2=0) / three bf16 terms
@pytest.mark.parametrize("n", [1, 40, 100, 256, 300, 700, 3000, 4096, 4100])
def test_bf16_fc_vs_independent_restatement(n, terms, model_of, orc):
    """DCE_BF16_FC (BASELINE configs[4]: the reference's fc layers, src/contact_cnn.py:47-58, with fc.0 / fc.3 on bf16
    operands) against oracle_forward_windows_bf16fc.  Stage by stage, each layer is checked on the DEVICE's own inputs
    of that layer, so a dropped K-tile, a mis-rounded conversion or a stale staging buffer cannot hide behind the
    loose end-to-end band:
      feat  : bf16 bits == RNE(fp32 features of the fp32 context)            (same conv kernel, only the store differs)
      h1    : bf16 bits == RNE(oracle fc.0 on the device's bf16 feat) up to 1 bf16 ulp on a few entries (fp32
              accumulation order vs fp64: a sum that lands within 1e-5 of a rounding boundary may round the other way)
      h2    : fp32 tolerance vs oracle fc.3 on the device's bf16 h1
      logits: fp32 tolerance vs oracle fc.6 on the device's h2
    and end to end against the restatement run from the windows."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    x = np.random.default_rng(11 + n).standard_normal((n, 150, 54), dtype=np.float32)
    if terms != "h2" and n not in (1, 300, 4096, 4100):
        pytest.skip("the options are covered at one size per FC kernel family")
    m16 = model_of(precision="bf16_fc", max_batch=8192, tune=None if terms == "h2" else {"bf16_conv_h2": 0} if terms == 2 else {"x3_bf16_terms": 3})
    m32 = model_of(max_batch=8192)
    nt = min(n, 512)                                               # taps on a bounded slice keep the CPU side in seconds
    sl = slice(n - nt, n)                                          # ... the LAST rows: partial tiles / the peeled remainder
    t16 = m16.forward_taps(x)
    feat32 = m32.forward_taps(x)["feat"]
    assert t16["feat"].dtype == np.uint16 and t16["h1"].dtype == np.uint16
    # (1) features.  The mode's conv stack runs on split bf16 operands at every batch size (csrc/conv_x3.hip; since round 4 from one
    # window up): values in another association than the fp32 context's, so a value close to a bf16 rounding boundary may round the
    # other way -- rare, and never by more than that.
    want16 = orc.bf16_round(feat32)
    got16 = orc.bf16_from_bits(t16["feat"])
    if not m16.last_plan()[0].startswith("conv_x"):              # (x3_conv=0 / x3_bf16_min: the fp32 context's conv kernel, only the store differs)
        assert np.array_equal(t16["feat"], orc.bf16_bits(want16))
    else:
        # (... by default on TWO-term operands, conv_x2_*: ~17 significant bits, so about one value in 2^8 sits close enough to a
        #  bf16 rounding boundary to land on its other side; the bound on each such value below is unchanged)
        assert m16.last_plan()[0].startswith(("conv_x2_bf16", "conv_x3_bf16")), m16.last_plan()
        fd = got16 != want16
        assert fd.mean() < (2e-2 if m16.last_plan()[0].startswith("conv_x2") else 2e-3), f"{fd.sum()} of {fd.size} features round differently"
        _, fe = np.frexp(np.maximum(np.abs(want16), np.abs(got16)))
        fbad = fd & (np.abs(got16 - feat32) > np.ldexp(0.5, fe - 8) + 1e-5 * np.abs(feat32).max() + 1e-4 * np.abs(feat32))
        assert not fbad.any(), f"{fbad.sum()} features differ by more than a rounding-boundary flip"
    # (2) fc.0 on the device's own bf16 features
    w1, w2 = orc.bf16_round(sd["fc.0.weight"]), orc.bf16_round(sd["fc.3.weight"])
    feat16 = orc.bf16_from_bits(t16["feat"][sl])
    h1_ref = orc.linear_rows(feat16, w1, sd["fc.0.bias"], relu=True)
    h1_dev = orc.bf16_from_bits(t16["h1"][sl])
    h1_ref16 = orc.bf16_round(h1_ref)
    diff = h1_dev != h1_ref16
    assert diff.mean() < 2e-3, f"{diff.sum()} of {diff.size} fc.0 outputs round differently"
    # a differently rounded entry must sit at a rounding boundary (or at ReLU's kink): the device's bf16 value is then
    # half a bf16 ulp (8 significand bits) from the fp64-accumulated sum, give or take the fp32 summation noise
    _, e = np.frexp(np.maximum(np.abs(h1_ref16), np.abs(h1_dev)))
    half_ulp = np.ldexp(0.5, e - 8)
    noise = 1e-5 * np.abs(h1_ref).max() + 1e-4 * np.abs(h1_ref)
    bad = diff & (np.abs(h1_dev - h1_ref) > half_ulp + noise)
    assert not bad.any(), f"{bad.sum()} fc.0 outputs differ by more than a rounding-boundary flip"
    # (3) fc.3 on the device's own bf16 h1, (4) fc.6 on the device's own h2
    h2_ref = orc.linear_rows(h1_dev, w2, sd["fc.3.bias"], relu=True)
    tol_ok(t16["h2"][sl], h2_ref, "bf16 fc.3 on the device's h1")
    lg_ref = orc.linear_rows(t16["h2"][sl], sd["fc.6.weight"], sd["fc.6.bias"], relu=False)
    tol_ok(t16["logits"][sl], lg_ref, "fc.6 on the device's h2")
    # (5) end to end: predict() is the same bits as the tap run, and stays within the accumulated rounding-boundary
    # band of the restatement (a few 1-ulp differences in feat / h1 move a logit by ~1e-4 of the logit scale)
    out = m16.predict(x)
    if terms == "h2": assert m16.last_plan()[0] == ("conv_h2_bf16_permk" if n > 256 else "conv_x2_bf16_permk"), m16.last_plan()
    ref = orc.Oracle(sd, bf16_fc=True).forward_windows(x[sl])
    scale = np.abs(ref["logits"]).max()
    if m16.last_plan()[0].endswith("_permk"):
        # from 128 windows predict() takes the features straight from the accumulators in the K order t' * 128 + c (the tap run
        # keeps the reference's flatten order): the same bf16 products, fc.0's fp32 accumulation in another order -- an h1 entry
        # at a bf16 rounding boundary may round the other way
        assert np.abs(out["logits"] - t16["logits"]).max() <= 2e-3 * scale
    else:
        assert np.array_equal(out["logits"], t16["logits"])
    err = np.abs(out["logits"][sl] - ref["logits"]).max()
    # (the band against THIS realisation of the mode: 3e-3 of the largest logit with the two-term conv stack -- worst over 1e6 windows
    #  2.6e-3, profiles/r4k_full_parity_1e6_bf16_fc.json -- 2e-3 with the three-term one)
    assert err <= (3e-3 if m16.last_plan()[0].startswith("conv_x2") else 2e-3) * scale, (err, scale)
    srt = np.sort(ref["logits"], axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 1e-2 * scale
    assert np.array_equal(out["pred"][sl][safe], ref["pred"][safe])
    assert np.array_equal(out["contacts"], orc.decimal2binary(out["pred"]))


def test_bf16_fc_online_and_sequence_paths_agree(model_of, orc):
    """The bf16-FC mode through the other entry points (z-score fused sequence call, online push) gives the rows the
    batch call gives: they share the kernels checked above."""
    from deep_contact_estimator_amd import synth
    m16 = model_of(precision="bf16_fc", max_batch=64)
    seq = synth.make_sequence(150 + 20, 31).astype(np.float32)
    out = m16.infer_sequence(seq)
    ref = orc.Oracle(synth.make_state_dict(1, "uniform"), bf16_fc=True).forward_windows(orc.zscore_windows(seq))
    scale = np.abs(ref["logits"]).max()
    assert np.abs(out["logits"] - ref["logits"]).max() <= 2e-3 * scale
    m16.online_reset()
    rows = [r for r in (m16.online_push(s) for s in seq) if r is not None]
    assert len(rows) == 21
    # the online push runs a one-window kernel sequence, the sequence call a 21-window one: different bf16 GEMM kernels,
    # hence the same tolerance, not bit equality (the fp32 mode IS bit-identical across batch sizes)
    assert np.abs(np.stack([r[0] for r in rows]) - ref["logits"]).max() <= 2e-3 * scale


# ------------------------------------------------------------------------------------------------
# packed rows
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("n", [1, 5, 70, 1030, 4096, 4100])
def test_packed_rows_equal_the_three_arrays(n, model_of):
    """dce_forward_windows_packed / dce_infer_sequence_packed write (n,68)-byte rows -- 16 fp32 logits + 4 contact
    bits -- from the SAME kernels (tail kernel, combine kernel, and both after a row cut): byte for byte what
    dce_forward_windows returns, on device and host pointers; dce_unpack_results inverts it."""
    import torch
    m = model_of()
    x = np.random.default_rng(5 + n).standard_normal((n, 150, 54), dtype=np.float32)
    ref = m.predict(x)
    want = np.concatenate([ref["logits"].view(np.uint8).reshape(n, 64), ref["contacts"]], axis=1)
    host = m.predict_packed(x)
    assert host.shape == (n, 68) and host.dtype == np.uint8 and np.array_equal(host, want)
    xd = torch.from_numpy(x).cuda()
    buf = torch.full((n, 68), 0xAB, dtype=torch.uint8, device="cuda")
    dev = m.predict_packed(xd, out=buf)
    assert dev.data_ptr() == buf.data_ptr() and np.array_equal(dev.cpu().numpy(), want)
    for packed in (host, dev):
        un = m.unpack_results(packed)
        un = {k: (v.cpu().numpy() if torch.is_tensor(v) else v) for k, v in un.items()}
        assert np.array_equal(un["logits"].view(np.uint32), ref["logits"].view(np.uint32))
        assert np.array_equal(un["pred"], ref["pred"]) and np.array_equal(un["contacts"], ref["contacts"])
    if n >= 70:
        seq = np.random.default_rng(n).standard_normal((n + 149, 54)).astype(np.float32)
        r2 = m.infer_sequence(seq)
        p2 = m.infer_sequence_packed(torch.from_numpy(seq).cuda()).cpu().numpy()
        assert np.array_equal(p2, np.concatenate([r2["logits"].view(np.uint8).reshape(n, 64), r2["contacts"]], axis=1))


# ------------------------------------------------------------------------------------------------
# the A/B switches are per context and select what they say
# ------------------------------------------------------------------------------------------------
def test_ab_switches_are_per_context(monkeypatch):
    """DESIGN.md's A/B switches belong to a CONTEXT (dce_create_ex's option string; round 2 latched them per process, which made
    every in-process A/B comparison run one kernel twice): live contexts created with different options run different kernels, as
    dce_last_plan reports, and still agree bit for bit; DCE_TUNE in the environment is the default of contexts created without
    options; an unknown key fails the creation."""
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    x = np.random.default_rng(1).standard_normal((4096, 150, 54), dtype=np.float32)
    from conftest import has_experiments
    scheds = ("tile", "phased", "lockstep") if has_experiments() else ("tile", "phased")
    made = {}
    for sched in scheds:
        if sched == "tile":                                  # ... through the environment: the default of a context without options
            monkeypatch.setenv("DCE_TUNE", "gemm_tile=1")
            made[sched] = contact_cnn(device=0, max_batch=4096); made[sched].load_state_dict(sd)
            made[sched].predict(x[:2])                       # creates the ctx under this environment
            monkeypatch.delenv("DCE_TUNE")
        else:
            made[sched] = contact_cnn(device=0, max_batch=4096, tune={"gemm_lockstep": int(sched == "lockstep")}); made[sched].load_state_dict(sd)
    if has_experiments():
        made["direct"] = contact_cnn(device=0, max_batch=4096, tune="conv_direct=1"); made["direct"].load_state_dict(sd)
    with pytest.raises(RuntimeError, match="unknown tuning option"):
        contact_cnn(device=0, max_batch=64, tune={"no_such_switch": 1})._ensure_ctx()
    outs, plans = {}, {}
    for k, m in made.items():                                # all four contexts alive, environment back to defaults
        outs[k] = m.predict(x)
        plans[k] = m.last_plan()
    assert plans["phased"][1] == "fc_phased256x128" and plans["phased"][2] == "fc23_fused_phased128x64", plans["phased"]
    assert plans["tile"][1] == "fc_tile128" and "phased" not in " ".join(plans["tile"]), plans["tile"]
    if "lockstep" in plans:
        assert plans["lockstep"][1] == "fc_lockstep256x128" and plans["lockstep"][2] == "fc23_fused_lockstep128x64", plans["lockstep"]
    assert plans["phased"][0] == "conv_wino2", plans["phased"]
    for k in scheds:
        assert np.array_equal(outs[k]["logits"], outs["phased"]["logits"]), k
    if "direct" in plans:
        assert plans["direct"][0] == "conv_direct", plans["direct"]
        tol_ok(outs["direct"]["logits"], outs["phased"]["logits"], "direct-form vs Winograd conv stack")
    for m in made.values():
        m.close()


@pytest.mark.experiments
def test_conv_rt4_equals_its_predecessor_bitwise(monkeypatch):
    """The two-window conv workgroup with four row tiles per wave (option conv4=1, an A/B variant) against the shipped
    two-row-tile kernel: same MFMAs per accumulator in the same K order -> the same feature bits, for pre-normalised windows
    and the z-score-fused sequence path, fp32 and bf16 features, odd window counts, NaN containment, repeated runs."""
    from deep_contact_estimator_amd import contact_cnn, synth
    sd = synth.make_state_dict(1, "uniform")
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2049, 150, 54), dtype=np.float32)
    x[7, 3, 5] = np.nan; x[1000, 100, 0] = np.inf
    seq = rng.standard_normal((1500 + 149, 54)).astype(np.float32) * 3 + 1
    for precision in ("fp32", "bf16_fc"):                   # (x3_conv=0: bf16_fc on the Winograd conv stack -- its default is conv_x3.hip)
        old = contact_cnn(device=0, max_batch=4096, precision=precision, tune={"x3_conv": 0}); old.load_state_dict(sd)
        ref_t = old.forward_taps(x); assert old.last_plan()[0] == "conv_wino2"
        ref_s = old.infer_sequence(seq)
        new = contact_cnn(device=0, max_batch=4096, precision=precision, tune={"x3_conv": 0, "conv4": 1}); new.load_state_dict(sd)
        for rep in range(3):
            t = new.forward_taps(x); assert new.last_plan()[0] == "conv_wino2_rt4"
            for k in ("feat", "logits"):
                assert np.array_equal(t[k].view(np.uint32 if t[k].dtype == np.float32 else np.uint16),
                                      ref_t[k].view(np.uint32 if t[k].dtype == np.float32 else np.uint16)), (precision, k, rep)
            s2 = new.infer_sequence(seq)
            assert np.array_equal(s2["logits"], ref_s["logits"]) and np.array_equal(s2["pred"], ref_s["pred"])
        assert np.isnan(t["logits"][7]).all() and np.isnan(t["logits"][1000]).all() and np.isfinite(t["logits"][8]).all()
        old.close(); new.close()


def test_plan_by_batch_size(model_of):
    """The kernel families the dispatch picks at the batch sizes the other tests rely on."""
    m = model_of()
    want = {1: ["conv_wino_quarter_ch2", "fc_gemv", "fc_gemv", "fc3_tail"],
            30: ["conv_wino_quarter_ch2", "fc_split16x16", "fc_split16x16", "fc3_tail"],
            100: ["conv_wino_half", "fc_chain32x32", "fc_chain32x32", "fc3_tail"],
            200: ["conv_wino1x8", "fc_chain32x32", "fc_chain32x32", "fc3_tail"]}
    rng = np.random.default_rng(2)
    for n, plan in want.items():
        m.predict(rng.standard_normal((n, 150, 54), dtype=np.float32))
        assert m.last_plan() == plan, (n, m.last_plan())
    m.predict(rng.standard_normal((4096, 150, 54), dtype=np.float32))
    p = m.last_plan()
    assert p[0] == "conv_wino2" and p[1:] == ["fc_phased256x128", "fc23_fused_phased128x64", "fc6_combine"], p


@pytest.mark.parametrize("n", [1, 8, 9, 16, 17, 30, 33, 64, 65, 96, 97, 200, 641, 700, 1030, 3000, 4096, 4100, 4130])
def test_fc_layers_bit_exact_against_the_summation_tree(n, model_of, orc):
    """Every fp32 FC kernel family -- GEMV (<= 32 windows), MFMA chain, 64x64 / 128x128 tiles, phased 128x64 / 256x128,
    the fused fc.3 + fc.6-chunk epilogue, and the row cuts that mix them -- returns, BIT FOR BIT, the fixed four-range
    summation tree of csrc/fc_tree.h evaluated on the CPU with fmaf (oracle_linear_rows_tree), on the device's own
    inputs of each layer: the "same bits at every batch size" property pinned on an independent model, not only on the
    kernels agreeing with each other."""
    from deep_contact_estimator_amd import synth
    sd = synth.make_state_dict(1, "uniform")
    m = model_of(max_batch=8192)
    x = np.random.default_rng(100 + n).standard_normal((n, 150, 54), dtype=np.float32)
    t = m.forward_taps(x)
    plan = m.last_plan()
    rows = np.unique(np.concatenate([np.arange(min(n, 40)), np.arange(max(n - 72, 0), n)]))     # first rows + the last (partial tiles, remainders)
    h1 = orc.linear_rows_tree(t["feat"][rows], sd["fc.0.weight"], sd["fc.0.bias"], relu=True)
    assert np.array_equal(h1.view(np.uint32), t["h1"][rows].view(np.uint32)), (n, plan, np.abs(h1 - t["h1"][rows]).max())
    h2 = orc.linear_rows_tree(t["h1"][rows], sd["fc.3.weight"], sd["fc.3.bias"], relu=True)
    assert np.array_equal(h2.view(np.uint32), t["h2"][rows].view(np.uint32)), (n, plan, np.abs(h2 - t["h2"][rows]).max())


def test_inference_and_compute_acc_reference_broadcast(golden, case_inputs, model_of, tmp_path):
    """reference_broadcast=True reproduces src/inference_one_seq.py:54 literally: with the (T,1) labels of
    mat2numpy_one_seq and batch_size 30 the reference's own function returns 1.8515625 (fixture loop_one_seq.npz,
    generated by running the reference) -- so does this one; at batch_size 1 both modes give the elementwise number."""
    from deep_contact_estimator_amd import synth
    from deep_contact_estimator_amd.data_handler import contact_dataset, WindowLoader
    from deep_contact_estimator_amd import inference as inf
    g = golden("loop_one_seq")
    seq64 = synth.make_sequence(int(g["T"]), int(g["sseed"]), str(g["kind"]))
    lab = synth.make_labels(int(g["T"]), int(g["sseed"]), two_d=True)
    np.save(tmp_path / "d.npy", seq64); np.save(tmp_path / "l.npy", lab)
    ds = contact_dataset(data_path=str(tmp_path / "d.npy"), label_path=str(tmp_path / "l.npy"), window_size=150, device="cuda")
    m = model_of(int(g["wseed"]), str(g["bias"]))
    for B in (1, 30):
        res, acc, leg = inf.inference_and_compute_acc(WindowLoader(ds, B), m, "cuda", reference_broadcast=True)
        assert np.array_equal(res.cpu().numpy(), g[f"contacts_B{B}"])
        assert acc == float(g[f"acc_B{B}"]), (B, acc)
        assert np.array_equal(leg, g[f"acc_per_leg_B{B}"])
    assert float(g["acc_B30"]) == 1.8515625

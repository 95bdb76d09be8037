import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_collection_modifyitems(config, items):
    """Tests marked `experiments` need the A/B variants: skipped unless the experiments library is the one loaded."""
    marked = [it for it in items if it.get_closest_marker("experiments")]
    if not marked:
        return
    try:
        ok = has_experiments()
    except Exception:                                      # no library built yet (CPU collection without a build)
        ok = False
    if not ok:
        skip = pytest.mark.skip(reason="needs libdce_experiments.so (tests/test_experiments_gpu.py runs these with DCE_LIB set)")
        for it in marked:
            it.add_marker(skip)


def pytest_configure(config):
    config.addinivalue_line("markers", "experiments: needs libdce_experiments.so (skipped with the product library)")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def tol_ok(got, ref, what="", factor=1.0):
    """The stated fp32 contract:
    |got-ref| <= 1e-5*max|ref| + 1e-4*|ref| elementwise (times `factor` where a test says why)."""
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    bound = factor * (1e-5 * np.abs(ref).max() + 1e-4 * np.abs(ref))
    err = np.abs(got - ref)
    bad = ~(err <= bound)           # also catches NaN
    assert not bad.any(), f"{what}: {bad.sum()} of {bad.size} outside tol; max err {np.nanmax(err):.3e} " \
                          f"(bound at worst {bound.reshape(-1)[np.nanargmax(err - bound)]:.3e})"


# tests/golden/chip_ar1.npz: the (window, class) logits of the REFERENCE that its own fp32 z-score (utils/data_handler.py:55-56) leaves
# just outside the fp32 contract of an fp64-statistics evaluation (1.001 .. 1.146 bounds; tests/test_oracle.py pins the list)
CHIP_AR1_REFERENCE_ZSCORE_OUTLIERS = [(384, 9), (386, 9), (580, 9), (2885, 7), (3584, 5)]


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return load


@pytest.fixture(scope="session")
def case_inputs():
    """Regenerate (state_dict, seq f32) for a golden case from its stored seeds and prove,
    by checksum, that they are the tensors the reference saw."""
    from deep_contact_estimator_amd import synth

    cache = {}

    def get(g):
        key = (int(g["wseed"]), str(g["bias"]), int(g["T"]), int(g["sseed"]), str(g["kind"]))
        if key not in cache:
            sd = synth.make_state_dict(key[0], key[1])
            seq = synth.make_sequence(key[2], key[3], key[4]).astype(np.float32)
            cs = np.array([float(seq.astype(np.float64).sum()), float(np.abs(seq.astype(np.float64)).sum())])
            assert np.array_equal(cs, g["seq_checksum"]), "synthetic sequence drifted from the golden run"
            for i, (k, _) in enumerate(synth.STATE_DICT_SHAPES):
                w = sd[k].astype(np.float64)
                assert np.array_equal(np.array([w.sum(), np.abs(w).sum()]), g["w_checksum"][i]), k
            cache[key] = (sd, seq)
        return cache[key]
    return get


def has_experiments() -> bool:
    """True when the loaded libdce is the experiments build (DCE_LIB=.../libdce_experiments.so): only there do the A/B variants
    exist (DCE_CONV4=1, DCE_GEMM=lockstep, DCE_X3_PAIR=1).  tests/test_experiments_gpu.py re-runs the tests that need it in a
    subprocess with that library; in the default run they skip."""
    from deep_contact_estimator_amd import _lib
    return _lib.has_experiments()


needs_experiments = pytest.mark.experiments

"""Deterministic synthetic checkpoints and sequences.

The reference's pretrained weights are a Google-Drive download (reference README.md:50-51)
and its dataset is external (README.md:9); neither exists offline.  Every test, golden
fixture and benchmark therefore regenerates weights/inputs from these seeded, torch-free
recipes: numpy's ``default_rng`` streams are version-stable, so the golden
generator (which imports the reference) and the GPU box (which cannot) get bit-identical
tensors.

PyTorch's default init is NOT used: with it every window collapses to one class with
~1e-3 top-2 margins, useless for argmax parity.  He-normal gives several classes and
O(1) margins.
"""
from __future__ import annotations

import numpy as np

# state_dict keys and shapes of contact_cnn (reference src/contact_cnn.py:8-58), in
# state_dict order.
STATE_DICT_SHAPES = (
    ("block1.0.weight", (64, 54, 3)), ("block1.0.bias", (64,)),
    ("block1.2.weight", (64, 64, 3)), ("block1.2.bias", (64,)),
    ("block2.0.weight", (128, 64, 3)), ("block2.0.bias", (128,)),
    ("block2.2.weight", (128, 128, 3)), ("block2.2.bias", (128,)),
    ("fc.0.weight", (2048, 4736)), ("fc.0.bias", (2048,)),
    ("fc.3.weight", (512, 2048)), ("fc.3.bias", (512,)),
    ("fc.6.weight", (16, 512)), ("fc.6.bias", (16,)),
)


def make_state_dict(seed: int = 1, bias: str = "uniform") -> dict[str, np.ndarray]:
    """He-normal weights; biases U(-0.1, 0.1) (``bias="uniform"``) or zero (``"zero"``)."""
    rng = np.random.default_rng(seed)
    sd: dict[str, np.ndarray] = {}
    for key, shape in STATE_DICT_SHAPES:
        if key.endswith(".weight"):
            fan_in = int(np.prod(shape[1:]))
            w = rng.standard_normal(shape, dtype=np.float32) * np.float32(np.sqrt(2.0 / fan_in))
            sd[key] = np.ascontiguousarray(w, dtype=np.float32)
        else:
            if bias == "zero":
                sd[key] = np.zeros(shape, dtype=np.float32)
            else:
                sd[key] = rng.uniform(-0.1, 0.1, size=shape).astype(np.float32)
    return sd


def make_sequence(T: int, seed: int = 0, kind: str = "normal", dtype=np.float64) -> np.ndarray:
    """(T,54) synthetic proprioceptive sequence.

    ``normal``: i.i.d. N(0,1).  ``ar1``: AR(1) rho=0.95 with per-channel offset U(-5,5) and
    scale 10**U(-2,1) -- stresses the z-score's cancellation.  float64 matches the on-disk
    dtype written by the reference's utils/mat2numpy.py:73,80.
    """
    rng = np.random.default_rng(seed)
    if kind == "normal":
        x = rng.standard_normal((T, 54))
    elif kind == "ar1":
        e = rng.standard_normal((T, 54))
        x = np.empty_like(e)
        x[0] = e[0]
        rho = 0.95
        s = np.sqrt(1.0 - rho * rho)
        for t in range(1, T):
            x[t] = rho * x[t - 1] + s * e[t]
        x = x * (10.0 ** rng.uniform(-2, 1, size=54)) + rng.uniform(-5, 5, size=54)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(x, dtype=dtype)


def make_labels(T: int, seed: int = 0, two_d: bool = False) -> np.ndarray:
    """Synthetic decimal contact labels in [0,16): (T,) like mat2numpy_split or (T,1) like
    mat2numpy_one_seq (reference utils/mat2numpy.py:76,175-177)."""
    lab = np.random.default_rng(seed + 7919).integers(0, 16, size=T).astype(np.int64)
    return lab.reshape(-1, 1) if two_d else lab

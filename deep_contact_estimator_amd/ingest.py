"""Input ingestion before the inference path -- SURVEY.md 8(f) rank 3 (host-side, no GPU work):

  binary2decimal        reference utils/mat2numpy.py:199-200   (T,4) contact bits -> decimal class id
  mat2numpy_one_seq     reference utils/mat2numpy.py:16-83     .mat -> (T,54) float64 .npy + (T,1) labels

The (T,54) column order is the contract the path depends on: q12, qd12, imu_acc3, imu_omega3, p12,
v12 (utils/mat2numpy.py:73; sliced back at src/inference_one_seq.py:74-79).  Legs are
[RF, LF, RH, LH], MSB first: [1,0,0,1] -> 9.
"""
from __future__ import annotations

import glob
import os

import numpy as np


def binary2decimal(a, axis: int = -1):
    a = np.asarray(a)
    return np.right_shift(np.packbits(a.astype(np.uint8), axis=axis), 8 - a.shape[axis]).squeeze()


def assemble(raw: dict):
    """One loaded .mat dict -> (data (T,54) float64, label (T,1))."""
    data = np.concatenate((raw["q"], raw["qd"], raw["imu_acc"], raw["imu_omega"], raw["p"], raw["v"]), axis=1)
    label = binary2decimal(raw["contacts"]).reshape((-1, 1))
    return data, label


def mat2numpy_one_seq(data_pth: str, save_pth: str):
    """Every .mat under data_pth* -> <save_pth><name>.npy and <name>_label.npy (no train/val/test split)."""
    import scipy.io as sio
    written = []
    for data_name in sorted(glob.glob(data_pth + "*")):
        data, label = assemble(sio.loadmat(data_name))
        stem = save_pth + os.path.splitext(os.path.basename(data_name))[0]
        np.save(stem + ".npy", data)
        np.save(stem + "_label.npy", label)
        written.append(stem + ".npy")
    return written

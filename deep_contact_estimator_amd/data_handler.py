"""Host-side mirror of the reference dataset (reference utils/data_handler.py:13-61).

``contact_dataset(data_path, label_path, window_size, device)`` keeps the reference's
constructor, ``__len__`` and ``__getitem__`` contract -- item i is the z-scored window of rows
[i, i+150) plus ``label[i+149]`` -- with the whole fp32 sequence resident in HBM (as the
reference keeps it on its device, :26-27).  The z-score itself runs in the HIP kernel
(dce_zscore_windows); ``WindowLoader`` is the batched equivalent of
``DataLoader(dataset, batch_size=B)`` (sequential sampler, default collate, short last batch):
one kernel launch per batch instead of ~5 tiny launches per window.

torch is used here only as the device-memory container (torch.Tensor on 'cuda').
"""
from __future__ import annotations

import numpy as np

from .contact_cnn import WINDOW, CHANNELS, contact_cnn, _device_index

_zs_models: dict[int, contact_cnn] = {}


def _zs_model(device) -> contact_cnn:
    """A weight-less context per device, used only for the z-score kernel."""
    idx = _device_index(device)
    if idx not in _zs_models:
        _zs_models[idx] = contact_cnn(device=idx, max_batch=1)
    return _zs_models[idx]


class contact_dataset:
    def __init__(self, data_path=None, label_path=None, window_size=WINDOW, device="cuda",
                 data=None, label=None):
        """data_path/label_path: the reference's .npy files ((T,54) float64 and (T,) or (T,1)
        integer labels, utils/mat2numpy.py:73-80,175-177).  ``data=`` / ``label=`` take arrays
        directly (synthetic runs)."""
        import torch
        if window_size != WINDOW:
            raise ValueError(f"window_size must be {WINDOW}: contact_cnn hard-codes 4736 = 128*37 features")
        if data is None:
            data = np.load(data_path)
        if label is None:
            label = np.load(label_path)          # required even without accuracy, as in the reference (:22)
        data = np.asarray(data)
        if data.ndim != 2 or data.shape[1] != CHANNELS:
            raise ValueError(f"data must be (T,{CHANNELS}), got {data.shape}")
        self.num_data = data.shape[0] - window_size + 1
        self.window_size = window_size
        self.device = torch.device(device)
        self.data = torch.from_numpy(np.ascontiguousarray(data)).type(torch.float32).to(self.device)
        self.label = torch.from_numpy(np.ascontiguousarray(label)).type(torch.int64).to(self.device)
        self._zs = _zs_model(self.device)

    def __len__(self):
        return self.num_data

    def windows(self, first: int, n: int):
        """z-scored windows [first, first+n) as a (n,150,54) device tensor."""
        return self._zs.zscore_windows(self.data, first, n)

    def __getitem__(self, idx):
        import torch
        if torch.is_tensor(idx):
            idx = idx.tolist()
        if idx < 0:
            idx += self.num_data
        if not 0 <= idx < self.num_data:
            raise IndexError(idx)
        this_data = self.windows(idx, 1)[0]
        this_label = self.label[idx + self.window_size - 1]
        return {"data": this_data, "label": this_label}


class WindowLoader:
    """DataLoader(dataset, batch_size=B) for contact_dataset: yields
    {'data': (b,150,54) f32, 'label': (b,) or (b,1) i64} in order, last batch short."""

    def __init__(self, dataset: contact_dataset, batch_size: int = 1):
        self.dataset = dataset
        self.batch_size = int(batch_size)

    def __len__(self):
        n = max(len(self.dataset), 0)
        return (n + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        ds = self.dataset
        n = max(len(ds), 0)
        for b0 in range(0, n, self.batch_size):
            b = min(self.batch_size, n - b0)
            lab = ds.label[b0 + ds.window_size - 1: b0 + ds.window_size - 1 + b]
            yield {"data": ds.windows(b0, b), "label": lab}

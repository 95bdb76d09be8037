"""PyTorch-CPU functional restatement of the reference path -- TEST INFRASTRUCTURE ONLY.

This is what bench.py times as ``cpu_baseline`` (kind "port"): the reference's own Python
cannot travel to the GPU box, so the same op sequence the reference dispatches on CPU
(MKL-DNN conv1d / MKL sgemm) is re-stated functionally here, from a plain state_dict.
tests/test_oracle.py checks it against the golden vectors captured from the imported
reference, so its timing is the reference's CPU timing up to Python-module overhead.

  z-score              utils/data_handler.py:55-56
  forward              src/contact_cnn.py:60-66
  argmax + bit unpack  src/inference_one_seq.py:26-27,59-62
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def to_torch(state_dict):
    return {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v), dtype=np.float32))
            for k, v in state_dict.items()}


@torch.no_grad()
def forward(sd, x):
    """x (B,150,54) f32 z-scored -> logits (B,16)."""
    x = x.permute(0, 2, 1)
    x = F.relu(F.conv1d(x, sd["block1.0.weight"], sd["block1.0.bias"], padding=1))
    x = F.relu(F.conv1d(x, sd["block1.2.weight"], sd["block1.2.bias"], padding=1))
    x = F.max_pool1d(x, 2, 2)
    x = F.relu(F.conv1d(x, sd["block2.0.weight"], sd["block2.0.bias"], padding=1))
    x = F.relu(F.conv1d(x, sd["block2.2.weight"], sd["block2.2.bias"], padding=1))
    x = F.max_pool1d(x, 2, 2)
    x = x.reshape(x.shape[0], -1)
    x = F.relu(F.linear(x, sd["fc.0.weight"], sd["fc.0.bias"]))
    x = F.relu(F.linear(x, sd["fc.3.weight"], sd["fc.3.bias"]))
    return F.linear(x, sd["fc.6.weight"], sd["fc.6.bias"])


def decimal2binary(x):
    mask = 2 ** torch.arange(3, -1, -1).to(x.device, x.dtype)
    return x.unsqueeze(-1).bitwise_and(mask).ne(0).byte()


@torch.no_grad()
def reference_loop(sd, seq, batch_size):
    """The reference's loop shape: per-item slice + z-score, stack, forward, argmax, unpack,
    cat (utils/data_handler.py:55-56, src/inference_one_seq.py:19-30).  seq: (T,54) f32 tensor."""
    n = seq.shape[0] - 149
    res = torch.empty(0, 4, dtype=torch.uint8)
    for b0 in range(0, n, batch_size):
        items = []
        for i in range(b0, min(b0 + batch_size, n)):
            w = seq[i:i + 150, :]
            items.append((w - torch.mean(w, dim=0)) / torch.std(w, dim=0))
        out = forward(sd, torch.stack(items))
        _, pred = torch.max(out, 1)
        res = torch.cat((res, decimal2binary(pred)), 0)
    return res

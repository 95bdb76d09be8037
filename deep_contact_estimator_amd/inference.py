"""Host-side mirror of the reference's inference/test loops.

  inference                   reference src/inference_one_seq.py:19-30
  inference_and_compute_acc   reference src/inference_one_seq.py:33-57
  compute_accuracy            reference src/test.py:72-107
  decimal2binary              reference src/inference_one_seq.py:59-62 (dup src/test.py:109-111)

The loops keep the reference's signatures and return types, but each iteration is ONE pass
through the C ABI (model.predict: logits + argmax + 4-bit unpack come out of the same kernel
sequence) and results land in pre-sized buffers instead of torch.cat / np.vstack regrowth.
``inference_sequence`` is the fused fast path (no materialised windows at all).
"""
from __future__ import annotations

import numpy as np


def decimal2binary(x):
    """class id(s) in [0,16) -> (...,4) uint8 contact bits, MSB first = legs [RF, LF, RH, LH]."""
    import torch
    if torch.is_tensor(x):
        mask = 2 ** torch.arange(4 - 1, -1, -1).to(x.device, x.dtype)
        return x.unsqueeze(-1).bitwise_and(mask).ne(0).byte()
    x = np.asarray(x)
    return ((x[..., None] & np.array([8, 4, 2, 1], dtype=x.dtype)) != 0).astype(np.uint8)


def inference(dataloader, model, device=None):
    """-> (N,4) uint8 tensor of contact states on the device (row j <-> data row j+149)."""
    import torch
    chunks = []
    for sample in dataloader:
        chunks.append(model.predict(sample["data"])["contacts"])
    if not chunks:
        return torch.empty(0, 4, dtype=torch.uint8, device=device)
    chunks = [c if torch.is_tensor(c) else torch.from_numpy(c) for c in chunks]
    return torch.cat(chunks, 0)


def inference_sequence(dataset, model):
    """Fused equivalent of inference(DataLoader(dataset, B), model) for any B: one
    dce_infer_sequence call over the device-resident sequence."""
    return model.infer_sequence(dataset.data)["contacts"]


def _counts(pred, contacts, gt_label):
    import torch
    gt = gt_label.reshape(-1).to(pred.device)
    bin_gt = decimal2binary(gt)
    per_leg = (contacts == bin_gt).sum(dim=0)
    correct = (pred.to(torch.int64) == gt).sum()
    return per_leg, correct, bin_gt


def inference_and_compute_acc(dataloader, model, device=None, reference_broadcast: bool = False):
    """-> (infer_results (N,4) u8, acc, acc_per_leg (4,)).  Labels may be (B,) or (B,1).  By default the class
    accuracy is elementwise for every batch size, which equals the reference at its shipped batch_size 1.
    reference_broadcast=True reproduces the reference LITERALLY: at src/inference_one_seq.py:54 it compares the
    (B,) predictions with the (B,1) labels mat2numpy_one_seq writes, which broadcasts to (B,B) and counts every
    (prediction j, label i) match of a batch -- a "class accuracy" that can exceed 1 at B>1.  With
    1-D labels the two modes agree."""
    import torch
    num_data = 0
    per_leg = None
    correct = None
    chunks = []
    for sample in dataloader:
        out = model.predict(sample["data"])
        pl, cr, _ = _counts(out["pred"], out["contacts"], sample["label"])
        if reference_broadcast:                                 # (prediction==gt_label).sum() with gt_label as the loader gave it
            cr = (out["pred"].to(torch.int64) == sample["label"].to(out["pred"].device)).sum()
        per_leg = pl if per_leg is None else per_leg + pl       # stays on the device: no per-batch sync
        correct = cr if correct is None else correct + cr
        num_data += out["pred"].shape[0]
        chunks.append(out["contacts"])
    if num_data == 0:
        return torch.empty(0, 4, dtype=torch.uint8, device=device), float("nan"), np.full(4, np.nan)
    return (torch.cat(chunks, 0), correct.item() / num_data,
            per_leg.cpu().numpy().astype(np.float64) / num_data)


def compute_accuracy(dataloader, model):
    """-> (acc, acc_per_leg(4), bin_pred_arr (N,4) f64, bin_gt_arr (N,4) f64, pred_arr (N,) f64,
    gt_arr (N,) f64) exactly as src/test.py:72-107 returns them (float64 numpy arrays)."""
    import torch
    num_data = 0
    per_leg = None
    correct = None
    preds, contacts, gts, bin_gts = [], [], [], []
    for sample in dataloader:
        out = model.predict(sample["data"])
        pl, cr, bin_gt = _counts(out["pred"], out["contacts"], sample["label"])
        per_leg = pl if per_leg is None else per_leg + pl
        correct = cr if correct is None else correct + cr
        num_data += out["pred"].shape[0]
        preds.append(out["pred"]); contacts.append(out["contacts"])
        gts.append(sample["label"].reshape(-1)); bin_gts.append(bin_gt)
    if num_data == 0:
        z = np.zeros((0, 4))
        return float("nan"), np.full(4, np.nan), z, z.copy(), np.zeros(0), np.zeros(0)
    cat = lambda xs: torch.cat(xs, 0).cpu().numpy().astype(np.float64)     # one D2H per array
    return (correct.item() / num_data, per_leg.cpu().numpy().astype(np.float64) / num_data,
            cat(contacts), cat(bin_gts), cat(preds), cat(gts))

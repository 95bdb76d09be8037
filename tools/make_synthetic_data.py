#!/usr/bin/env python3
"""Write the synthetic stand-ins for the reference's external artefacts (dataset .npy files and
the pretrained checkpoint are not available offline): float64 (T,54) data, decimal labels, and a
checkpoint with the reference's torch schema {'model_state_dict': ...} (src/train.py:145-153).

    python tools/make_synthetic_data.py [--out synthetic_data] [--T 1149]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deep_contact_estimator_amd import synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="synthetic_data")
    ap.add_argument("--T", type=int, default=150 + 999)
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    np.save(os.path.join(a.out, "one_seq_data.npy"), synth.make_sequence(a.T, 0))
    np.save(os.path.join(a.out, "one_seq_label.npy"), synth.make_labels(a.T, 0, two_d=True))
    np.save(os.path.join(a.out, "test.npy"), synth.make_sequence(a.T, 1))
    np.save(os.path.join(a.out, "test_label.npy"), synth.make_labels(a.T, 1))
    sd = synth.make_state_dict(1, "uniform")
    try:
        import torch
        torch.save({"epoch": 0, "model_state_dict": {k: torch.from_numpy(v) for k, v in sd.items()}},
                   os.path.join(a.out, "synthetic_contact_cnn.pt"))
    except ImportError:
        np.savez(os.path.join(a.out, "synthetic_contact_cnn.npz"), **sd)
    print("wrote", sorted(os.listdir(a.out)))


if __name__ == "__main__":
    main()

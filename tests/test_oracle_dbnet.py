"""CPU: the oracle restatement reproduces the golden vector written by the REFERENCE's DBNet
class (oracle/pin_against_reference.py) - this is what pins the oracle."""
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(__file__), "golden", "dbnet_ref_64x96.npz")


def test_oracle_matches_reference_golden():
    from oracle.dbnet import dbnet_forward
    from yomitoku_amd.utils.synth import dbnet_state_dict

    z = np.load(GOLD)
    sd = dbnet_state_dict(int(z["seed"]))
    out = dbnet_forward(sd, torch.from_numpy(z["x"]))["binary"].numpy()
    assert out.shape == z["prob"].shape
    assert np.abs(out - z["prob"]).max() < 1e-6


def test_synthetic_checkpoint_is_deterministic_and_complete():
    from yomitoku_amd.utils.synth import dbnet_state_dict

    a, b = dbnet_state_dict(7), dbnet_state_dict(7)
    assert list(a) == list(b)
    assert all(torch.equal(a[k], b[k]) for k in a)
    # torchvision resnet50 + decoder parameter count (25.56 M)
    n = sum(v.numel() for k, v in a.items() if "num_batches" not in k and "running" not in k)
    assert 25.4e6 < n < 25.7e6

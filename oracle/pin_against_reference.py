"""ORACLE tooling: pin the oracle restatements against the reference's own code (run in the
build container where /root/reference exists) and write the golden vectors under tests/golden/.

    python -m oracle.pin_against_reference [dbnet|parseq|rtdetr|all]

Every golden file records the seed of the synthetic checkpoint (yomitoku_amd/utils/synth.py), the
input, and the output of the REFERENCE implementation; tests compare both the oracle (CPU) and the
HIP path (GPU box) against it.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from ._refstubs import AttrDict, ref_import

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def pin_dbnet():
    from yomitoku_amd.utils.synth import dbnet_state_dict

    from .dbnet import dbnet_forward

    mod = ref_import("yomitoku.models.dbnet_plus")
    cfg = AttrDict(
        backbone={"name": "resnet50", "dilation": True},
        decoder={"in_channels": [256, 512, 1024, 2048], "hidden_dim": 256, "adaptive": True, "serial": True,
                 "smooth": False, "k": 50},
    )
    seed = 1234
    sd = dbnet_state_dict(seed)
    model = mod.DBNet(cfg)
    missing = model.load_state_dict(sd, strict=True)  # key names + shapes must match the reference exactly
    model.eval()
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(64))
    with torch.inference_mode():
        ref = model(x)["binary"]
    ours = dbnet_forward(sd, x)["binary"]
    err = (ref - ours).abs().max().item()
    print(f"[dbnet] reference vs oracle: max abs diff {err:.3e} ({missing})")
    assert err < 1e-6, err
    np.savez_compressed(os.path.join(GOLDEN, "dbnet_ref_64x96.npz"), seed=seed, x=x.numpy(), prob=ref.numpy())


def main(argv):
    what = argv[1] if len(argv) > 1 else "all"
    os.makedirs(GOLDEN, exist_ok=True)
    todo = {"dbnet": pin_dbnet}
    for k, fn in todo.items():
        if what in (k, "all"):
            fn()


if __name__ == "__main__":
    main(sys.argv)

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


def pin_parseq():
    from types import SimpleNamespace

    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    from .parseq import PRESETS, make_cfg, parseq_forward

    mod = ref_import("yomitoku.models.parseq")
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    rcfg = AttrDict(
        max_label_length=100, decode_ar=1, refine_iters=1, num_tokens=ocfg.num_tokens,
        data={"img_size": [32, 800]},
        encoder={"patch_size": [4, 8], "num_heads": 6, "embed_dim": 192, "mlp_ratio": 4, "depth": 12},
        decoder={"embed_dim": 192, "num_heads": 6, "mlp_ratio": 4, "depth": 1},
    )
    cases = [
        # (file tag, checkpoint kwargs, batch, width)
        ("eos", dict(seed=1235, eos_bias=4.5), 3, 96),
        ("rep", dict(seed=1236, eos_bias=3.0, favour_token=17, favour_bias=12.0), 3, 72),
    ]
    for tag, kw, bs, width in cases:
        sd = parseq_state_dict(**kw)
        model = mod.PARSeq(rcfg)
        res = model.load_state_dict(sd, strict=True)
        model.eval()
        model.tokenizer = SimpleNamespace(eos_id=0, bos_id=ocfg.num_tokens - 2, pad_id=ocfg.num_tokens - 1)
        x = synthetic_line_batch(11, bs, width)
        with torch.inference_mode():
            ref = model(x)
        ours, steps = parseq_forward(sd, ocfg, x, return_steps=True)
        err = (ref - ours).abs().max().item()
        ids = ref.argmax(-1)
        lens = [int((row == 0).nonzero()[0]) if (row == 0).any() else len(row) for row in ids]
        print(f"[parseq/{tag}] reference vs oracle: max abs diff {err:.3e}, AR steps {steps}, lengths {lens} ({res})")
        assert err < 1e-5, err
        # logits are large (B x 101 x 7119): keep the decisive slices - arg-max ids, max logit, a strided sample
        np.savez_compressed(
            os.path.join(GOLDEN, f"parseq_ref_{tag}.npz"), ckpt=np.array(repr(kw)), x=x.numpy(), steps=steps,
            ids=ids.numpy().astype(np.int32), top=ref.max(-1).values.numpy(), sample=ref[:, :, ::97].numpy(),
        )


def main(argv):
    what = argv[1] if len(argv) > 1 else "all"
    os.makedirs(GOLDEN, exist_ok=True)
    todo = {"dbnet": pin_dbnet, "parseq": pin_parseq}
    for k, fn in todo.items():
        if what in (k, "all"):
            fn()


if __name__ == "__main__":
    main(sys.argv)

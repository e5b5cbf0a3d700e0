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


def pin_rtdetr():
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    from .rtdetr import rtdetr_forward

    mod = ref_import("yomitoku.models.rtdetr")
    import omegaconf  # the stub: RTDETRTransformerv2 wants num_points as a ListConfig

    for tag, nc, seed in (("layout", 6, 1240), ("table", 3, 1241)):
        cfg = AttrDict(
            PResNet={"depth": 50, "variant": "d", "freeze_at": 0, "return_idx": [1, 2, 3], "num_stages": 4,
                     "freeze_norm": True},
            HybridEncoder={"in_channels": [512, 1024, 2048], "feat_strides": [8, 16, 32], "hidden_dim": 256,
                           "use_encoder_idx": [2], "num_encoder_layers": 1, "nhead": 8, "dim_feedforward": 1024,
                           "dropout": 0.0, "enc_act": "gelu", "expansion": 1.0, "depth_mult": 1, "act": "silu"},
            RTDETRTransformerv2={"num_classes": nc, "feat_channels": [256, 256, 256], "feat_strides": [8, 16, 32],
                                 "hidden_dim": 256, "num_levels": 3, "num_layers": 6, "num_queries": 300,
                                 "num_denoising": 100, "label_noise_ratio": 0.5, "box_noise_scale": 1.0,
                                 "eval_spatial_size": [640, 640], "eval_idx": -1,
                                 "num_points": omegaconf.ListConfig([4, 4, 4]), "cross_attn_method": "default",
                                 "query_select_method": "default"},
        )
        sd = rtdetr_state_dict(seed, num_classes=nc)
        model = mod.RTDETRv2(cfg)
        res = model.load_state_dict(sd, strict=True)
        model.eval()
        x = torch.rand(1, 3, 640, 640, generator=torch.Generator().manual_seed(seed))
        with torch.inference_mode():
            ref = model(x)
        ours = rtdetr_forward(sd, x)
        e1 = (ref["pred_logits"] - ours["pred_logits"]).abs().max().item()
        e2 = (ref["pred_boxes"] - ours["pred_boxes"]).abs().max().item()
        sc = ref["pred_logits"].sigmoid()
        print(f"[rtdetr/{tag}] reference vs oracle: logits {e1:.3e} boxes {e2:.3e}; scores>0.5: {(sc > 0.5).sum().item()} ({res})")
        assert e1 < 1e-5 and e2 < 1e-6, (e1, e2)
        np.savez_compressed(os.path.join(GOLDEN, f"rtdetr_ref_{tag}.npz"), seed=seed, num_classes=nc,
                            x_seed=seed, logits=ref["pred_logits"].numpy(), boxes=ref["pred_boxes"].numpy())


def _random_layout(rng, n, page=(1200, 1600)):
    """Boxes that look like page elements: columns of stacked blocks plus a few strays/overlaps."""
    boxes = []
    ncol = int(rng.integers(1, 4))
    col_w = page[0] // ncol
    for c in range(ncol):
        y = int(rng.integers(20, 120))
        while y < page[1] - 80 and len(boxes) < n:
            h = int(rng.integers(20, 260))
            x1 = c * col_w + int(rng.integers(5, 60))
            x2 = (c + 1) * col_w - int(rng.integers(5, 60))
            if rng.random() < 0.25:  # short block
                x2 = x1 + int(rng.integers(40, max(41, (x2 - x1) // 2)))
            boxes.append([x1, y, max(x1 + 10, x2), y + h])
            y += h + int(rng.integers(-10, 60))
    while len(boxes) < n:
        x1, y1 = int(rng.integers(0, page[0] - 50)), int(rng.integers(0, page[1] - 50))
        boxes.append([x1, y1, x1 + int(rng.integers(10, 400)), y1 + int(rng.integers(10, 200))])
    rng.shuffle(boxes)
    return [list(map(int, b)) for b in boxes[:n]]


def pin_host_logic():
    """Golden answers of the reference's pure-Python stages (reading order, containment filters,
    table cell grid) on seeded random layouts -> tests/golden/host_logic.json."""
    import json
    from types import SimpleNamespace

    ro = ref_import("yomitoku.reading_order")
    misc = ref_import("yomitoku.utils.misc")

    class El(SimpleNamespace):
        def dict(self):
            return {"box": self.box, "order": self.order}

    rng = np.random.default_rng(2024)
    cases = []
    for k in range(240):
        n = int(rng.integers(2, 40))
        boxes = _random_layout(rng, n)
        direction = ["top2bottom", "right2left", "left2right"][k % 3]
        els = [El(box=b, order=0) for b in boxes]
        ro.prediction_reading_order(els, direction)
        cases.append({"direction": direction, "boxes": boxes, "order": [int(e.order) for e in els]})
    pairs = []
    for _ in range(300):
        a = _random_layout(rng, 1)[0]
        b = [a[0] + int(rng.integers(-30, 30)), a[1] + int(rng.integers(-30, 30)), a[2] + int(rng.integers(-30, 30)),
             a[3] + int(rng.integers(-30, 30))]
        if b[2] <= b[0] or b[3] <= b[1]:
            continue
        ratio, inter = misc.calc_overlap_ratio(a, b)
        pairs.append({"a": a, "b": b, "ratio": float(ratio), "inter": inter, "contained": bool(misc.is_contained(a, b)),
                      "ih": bool(misc.is_intersected_horizontal(a, b)), "iv": bool(misc.is_intersected_vertical(a, b))})
    with open(os.path.join(GOLDEN, "host_logic.json"), "w") as f:
        json.dump({"reading_order": cases, "pairs": pairs}, f)
    print(f"[host] wrote {len(cases)} reading-order cases, {len(pairs)} box pairs")


def main(argv):
    what = argv[1] if len(argv) > 1 else "all"
    os.makedirs(GOLDEN, exist_ok=True)
    todo = {"dbnet": pin_dbnet, "parseq": pin_parseq, "rtdetr": pin_rtdetr, "host": pin_host_logic}
    for k, fn in todo.items():
        if what in (k, "all"):
            fn()


if __name__ == "__main__":
    main(sys.argv)

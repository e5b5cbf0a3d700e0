"""Seeded synthetic checkpoint for RT-DETRv2 (layout parser / table structure recogniser) with the
reference's parameter names (models/rtdetr.py, models/layers/rtdetr_*.py), plus the
two input-independent tables the reference builds in Python: the logit-space anchors / valid mask
(rtdetrv2_decoder.py:662-693) and the AIFI sin-cos position embedding
(rtdetr_hybrid_encoder.py:346-363)."""

from __future__ import annotations

import math
from collections import OrderedDict

import torch

from .synth import _Draw


def generate_anchors(eval_size=(640, 640), strides=(8, 16, 32), grid_size=0.05, eps=1e-2):
    """Anchors in logit space + validity mask, one row per memory token, levels concatenated."""
    anchors = []
    for lvl, s in enumerate(strides):
        h, w = int(eval_size[0] / s), int(eval_size[1] / s)
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
        xy = (torch.stack([gx, gy], dim=-1).unsqueeze(0) + 0.5) / torch.tensor([w, h], dtype=torch.float32)
        wh = torch.ones_like(xy) * grid_size * (2.0**lvl)
        anchors.append(torch.concat([xy, wh], dim=-1).reshape(-1, h * w, 4))
    anchors = torch.concat(anchors, dim=1)
    valid = ((anchors > eps) * (anchors < 1 - eps)).all(-1, keepdim=True)
    anchors = torch.log(anchors / (1 - anchors))
    anchors = torch.where(valid, anchors, torch.inf)
    return anchors, valid


def sincos_pos_embed(w: int, h: int, dim: int = 256, temperature: float = 10000.0):
    """[1, w*h, dim] table exactly as the reference lays it out (token t <-> (t // h, t % h))."""
    gw, gh = torch.meshgrid(torch.arange(int(w), dtype=torch.float32), torch.arange(int(h), dtype=torch.float32),
                            indexing="ij")
    pos_dim = dim // 4
    omega = torch.arange(pos_dim, dtype=torch.float32) / pos_dim
    omega = 1.0 / (temperature**omega)
    ow = gw.flatten()[..., None] @ omega[None]
    oh = gh.flatten()[..., None] @ omega[None]
    return torch.concat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)[None]


def rtdetr_state_dict(seed: int = 1240, num_classes: int = 6, hidden: int = 256, num_layers: int = 6, ffn: int = 1024,
                      num_queries: int = 300, score_bias: float = 0.0, score_gain: float = 1.0,
                      eval_size=(640, 640), enc_score_gain: float = 1.0) -> "OrderedDict[str, torch.Tensor]":
    d = _Draw(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def convnorm(name, cout, cin, k, gamma=(0.8, 1.2), gain=2.0, tracked=False):
        sd[name + ".conv.weight"] = d.normal((cout, cin, k, k), std=math.sqrt(gain / (cin * k * k)))
        sd[name + ".norm.weight"] = d.uniform((cout,), *gamma)
        sd[name + ".norm.bias"] = d.normal((cout,), std=0.05)
        sd[name + ".norm.running_mean"] = d.normal((cout,), std=0.05)
        sd[name + ".norm.running_var"] = d.uniform((cout,), 0.8, 1.2)
        if tracked:
            sd[name + ".norm.num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)

    def linear(name, out, inp, gain=1.0, bias_std=0.02):
        sd[name + ".weight"] = d.normal((out, inp), std=gain / math.sqrt(inp))
        sd[name + ".bias"] = d.normal((out,), std=bias_std)

    def lnorm(name, c):
        sd[name + ".weight"] = d.uniform((c,), 0.8, 1.2)
        sd[name + ".bias"] = d.normal((c,), std=0.05)

    def mha(name, c):
        sd[name + ".in_proj_weight"] = d.normal((3 * c, c), std=1.5 / math.sqrt(c))
        sd[name + ".in_proj_bias"] = d.normal((3 * c,), std=0.02)
        linear(name + ".out_proj", c, c, gain=0.7)

    # ---- PResNet-50vd, FrozenBatchNorm (no num_batches_tracked)
    b = "backbone."
    for name, cin, cout in (("conv1_1", 3, 32), ("conv1_2", 32, 32), ("conv1_3", 32, 64)):
        convnorm(b + "conv1." + name, cout, cin, 3)
    ch_in = 64
    for s, (ch, count) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3))):
        for i in range(count):
            p = f"{b}res_layers.{s}.blocks.{i}."
            convnorm(p + "branch2a", ch, ch_in, 1)
            convnorm(p + "branch2b", ch, ch, 3)
            convnorm(p + "branch2c", ch * 4, ch, 1, gamma=(0.2, 0.4))
            if i == 0:
                short = p + ("short" if s == 0 else "short.conv")
                convnorm(short, ch * 4, ch_in, 1, gain=1.0)
                ch_in = ch * 4
    # ---- HybridEncoder
    e = "encoder."
    for i, cin in enumerate((512, 1024, 2048)):
        convnorm(f"{e}input_proj.{i}", hidden, cin, 1, gain=1.0, tracked=True)
    a = e + "encoder.0.layers.0."
    mha(a + "self_attn", hidden)
    linear(a + "linear1", ffn, hidden)
    linear(a + "linear2", hidden, ffn, gain=0.5)
    lnorm(a + "norm1", hidden)
    lnorm(a + "norm2", hidden)

    def csp(name):
        convnorm(name + ".conv1", hidden, 2 * hidden, 1, tracked=True)
        convnorm(name + ".conv2", hidden, 2 * hidden, 1, tracked=True)
        for j in range(3):
            convnorm(f"{name}.bottlenecks.{j}.conv1", hidden, hidden, 3, tracked=True, gain=1.5)
            convnorm(f"{name}.bottlenecks.{j}.conv2", hidden, hidden, 1, tracked=True, gain=0.5)

    for i in range(2):
        convnorm(f"{e}lateral_convs.{i}", hidden, hidden, 1, tracked=True)
        csp(f"{e}fpn_blocks.{i}")
    for i in range(2):
        convnorm(f"{e}downsample_convs.{i}", hidden, hidden, 3, tracked=True)
        csp(f"{e}pan_blocks.{i}")
    # ---- RTDETRTransformerv2
    t = "decoder."
    for i in range(3):
        convnorm(f"{t}input_proj.{i}", hidden, hidden, 1, gain=1.0, tracked=True)
    for i in range(num_layers):
        p = f"{t}decoder.layers.{i}."
        mha(p + "self_attn", hidden)
        lnorm(p + "norm1", hidden)
        linear(p + "cross_attn.sampling_offsets", 8 * 12 * 2, hidden, gain=0.5, bias_std=1.0)
        linear(p + "cross_attn.attention_weights", 8 * 12, hidden, gain=1.0)
        linear(p + "cross_attn.value_proj", hidden, hidden)
        linear(p + "cross_attn.output_proj", hidden, hidden, gain=0.7)
        sd[p + "cross_attn.num_points_scale"] = torch.full((12,), 0.25)
        lnorm(p + "norm2", hidden)
        linear(p + "linear1", ffn, hidden)
        linear(p + "linear2", hidden, ffn, gain=0.5)
        lnorm(p + "norm3", hidden)
    sd[t + "denoising_class_embed.weight"] = d.normal((num_classes + 1, hidden), std=0.02)
    linear(t + "query_pos_head.layers.0", 2 * hidden, 4)
    linear(t + "query_pos_head.layers.1", hidden, 2 * hidden, gain=0.5)
    linear(t + "enc_output.proj", hidden, hidden)
    lnorm(t + "enc_output.norm", hidden)
    linear(t + "enc_score_head", num_classes, hidden, gain=1.0, bias_std=0.5)

    def mlp3(name):
        linear(name + ".layers.0", hidden, hidden)
        linear(name + ".layers.1", hidden, hidden)
        linear(name + ".layers.2", 4, hidden, gain=0.3)

    mlp3(t + "enc_bbox_head")
    for i in range(num_layers):
        linear(f"{t}dec_score_head.{i}", num_classes, hidden, gain=1.5, bias_std=0.5)
    for i in range(num_layers):
        mlp3(f"{t}dec_bbox_head.{i}")
    if score_bias or score_gain != 1.0:
        # widen / shift the final class logits so that only a realistic handful of the 300 queries clear
        # the score threshold (random heads would otherwise "detect" hundreds of boxes per page, or none)
        sd[f"{t}dec_score_head.{num_layers - 1}.weight"] *= score_gain
        sd[f"{t}dec_score_head.{num_layers - 1}.bias"] += score_bias
    if enc_score_gain != 1.0:
        # spread the encoder-token scores: with 18900 tokens (960 x 960) the top-1500 cut otherwise falls between two
        # scores ~1e-6 apart, and WHICH token is query no. 1500 would depend on fp32 summation order
        sd[t + "enc_score_head.weight"] *= enc_score_gain
    anchors, valid = generate_anchors(tuple(eval_size))  # 640 x 640: layout / table structure; 960 x 960: cell detector
    sd[t + "anchors"] = anchors
    sd[t + "valid_mask"] = valid
    return sd

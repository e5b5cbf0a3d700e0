"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference RT-DETRv2 (layout parser / table structure recogniser network) as
functions of a state dict, plain PyTorch fp32.  Paths relative to /root/reference/src/yomitoku.
Everything on this path is the reference's own Python + torch (no third-party arithmetic), and the
three layer files import unmodified in the build container, so oracle/pin_against_reference.py pins
this file against the real `RTDETRv2` class end to end (golden vectors in tests/golden/).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

_STAGES = ((64, 3), (128, 4), (256, 6), (512, 3))


def _cn(sd, x, name, stride=1, padding=0, act=None):
    """ConvNormLayer (rtdetr_backbone.py:27-56 / rtdetr_hybrid_encoder.py:17-46); FrozenBatchNorm2d
    (:177-226) and eval BatchNorm2d share y = x * w / sqrt(rv + 1e-5) + (b - rm * scale)."""
    y = F.conv2d(x, sd[name + ".conv.weight"], None, stride, padding)
    w, b = sd[name + ".norm.weight"], sd[name + ".norm.bias"]
    rm, rv = sd[name + ".norm.running_mean"], sd[name + ".norm.running_var"]
    scale = w * (rv + 1e-5).rsqrt()
    y = y * scale.reshape(1, -1, 1, 1) + (b - rm * scale).reshape(1, -1, 1, 1)
    if act == "relu":
        y = F.relu(y)
    elif act == "silu":
        y = F.silu(y)
    return y


def presnet(sd, x, prefix="backbone."):
    """rtdetr_backbone.py:245-334: ResNet-50-vd, return_idx [1, 2, 3]."""
    for n, s in (("conv1_1", 2), ("conv1_2", 1), ("conv1_3", 1)):
        x = _cn(sd, x, f"{prefix}conv1.{n}", s, 1, "relu")
    x = F.max_pool2d(x, 3, 2, 1)
    outs = []
    for s, (ch, count) in enumerate(_STAGES):
        for i in range(count):
            p = f"{prefix}res_layers.{s}.blocks.{i}."
            stride = 2 if (i == 0 and s != 0) else 1
            y = _cn(sd, x, p + "branch2a", 1, 0, "relu")
            y = _cn(sd, y, p + "branch2b", stride, 1, "relu")
            y = _cn(sd, y, p + "branch2c")
            if i == 0:
                if stride == 2:  # variant d: avgpool(2, 2, ceil) then 1x1
                    short = _cn(sd, F.avg_pool2d(x, 2, 2, 0, ceil_mode=True), p + "short.conv")
                else:
                    short = _cn(sd, x, p + "short")
            else:
                short = x
            x = F.relu(y + short)
        if s >= 1:
            outs.append(x)
    return outs


def _mha(sd, prefix, heads, q, k, v):
    out, _ = F.multi_head_attention_forward(
        q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), q.shape[-1], heads, sd[prefix + "in_proj_weight"],
        sd[prefix + "in_proj_bias"], None, None, False, 0.0, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"],
        training=False, need_weights=True,
    )
    return out.transpose(0, 1)


def _lin(sd, x, name):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def _ln(sd, x, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def _sincos(w, h, dim=256, temperature=10000.0):
    # rtdetr_hybrid_encoder.py:346-363
    gw, gh = torch.meshgrid(torch.arange(int(w), dtype=torch.float32), torch.arange(int(h), dtype=torch.float32),
                            indexing="ij")
    pos_dim = dim // 4
    omega = torch.arange(pos_dim, dtype=torch.float32) / pos_dim
    omega = 1.0 / (temperature**omega)
    ow = gw.flatten()[..., None] @ omega[None]
    oh = gh.flatten()[..., None] @ omega[None]
    return torch.concat([ow.sin(), ow.cos(), oh.sin(), oh.cos()], dim=1)[None]


def _csp(sd, x, name):
    # CSPRepLayer (rtdetr_hybrid_encoder.py:178-213) with un-fused RepVggBlocks (:124-141); conv3 = Identity
    x1 = _cn(sd, x, name + ".conv1", act="silu")
    for j in range(3):
        b = f"{name}.bottlenecks.{j}"
        x1 = F.silu(_cn(sd, x1, b + ".conv1", 1, 1) + _cn(sd, x1, b + ".conv2"))
    return x1 + _cn(sd, x, name + ".conv2", act="silu")


def hybrid_encoder(sd, feats, prefix="encoder."):
    """rtdetr_hybrid_encoder.py:365-414."""
    proj = [_cn(sd, f, f"{prefix}input_proj.{i}") for i, f in enumerate(feats)]
    h, w = proj[2].shape[2:]
    src = proj[2].flatten(2).permute(0, 2, 1)
    pos = _sincos(w, h)
    a = prefix + "encoder.0.layers.0."
    q = src + pos
    src = _ln(sd, src + _mha(sd, a + "self_attn.", 8, q, q, src), a + "norm1")
    src = _ln(sd, src + _lin(sd, F.gelu(_lin(sd, src, a + "linear1")), a + "linear2"), a + "norm2")
    proj[2] = src.permute(0, 2, 1).reshape(-1, 256, h, w).contiguous()
    inner = [proj[2]]
    for idx in (2, 1):
        high = _cn(sd, inner[0], f"{prefix}lateral_convs.{2 - idx}", act="silu")
        inner[0] = high
        up = F.interpolate(high, scale_factor=2.0, mode="nearest")
        inner.insert(0, _csp(sd, torch.concat([up, proj[idx - 1]], dim=1), f"{prefix}fpn_blocks.{2 - idx}"))
    outs = [inner[0]]
    for idx in range(2):
        down = _cn(sd, outs[-1], f"{prefix}downsample_convs.{idx}", 2, 1, "silu")
        outs.append(_csp(sd, torch.concat([down, inner[idx + 1]], dim=1), f"{prefix}pan_blocks.{idx}"))
    return outs


def _mlp(sd, x, name, n):
    for i in range(n):
        x = _lin(sd, x, f"{name}.layers.{i}")
        if i < n - 1:
            x = F.relu(x)
    return x


def _inverse_sigmoid(x, eps=1e-5):
    x = x.clip(min=0.0, max=1.0)
    return torch.log(x.clip(min=eps) / (1 - x).clip(min=eps))


def _deform_attn(sd, name, query, ref, value, shapes):
    """MSDeformableAttention.forward (rtdetrv2_decoder.py:141-219, 4-d reference points) +
    deformable_attention_core_func_v2 (:306-388, method "default")."""
    bs, lq = query.shape[:2]
    nh, npts, hd = 8, 12, 32
    value = _lin(sd, value, name + "value_proj").reshape(bs, -1, nh, hd)
    off = _lin(sd, query, name + "sampling_offsets").reshape(bs, lq, nh, npts, 2)
    aw = F.softmax(_lin(sd, query, name + "attention_weights").reshape(bs, lq, nh, npts), dim=-1)
    nps = sd[name + "num_points_scale"].unsqueeze(-1)
    loc = ref[:, :, None, :, :2] + off * nps * ref[:, :, None, :, 2:] * 0.5
    split = [h * w for h, w in shapes]
    vlist = value.permute(0, 2, 3, 1).flatten(0, 1).split(split, dim=-1)
    grids = (2 * loc - 1).permute(0, 2, 1, 3, 4).flatten(0, 1).split([4, 4, 4], dim=-2)
    sampled = []
    for lvl, (h, w) in enumerate(shapes):
        sampled.append(F.grid_sample(vlist[lvl].reshape(bs * nh, hd, h, w), grids[lvl], mode="bilinear",
                                     padding_mode="zeros", align_corners=False))
    aw = aw.permute(0, 2, 1, 3).reshape(bs * nh, 1, lq, npts)
    out = (torch.concat(sampled, dim=-1) * aw).sum(-1).reshape(bs, nh * hd, lq).permute(0, 2, 1)
    return _lin(sd, out, name + "output_proj")


def rtdetr_decoder(sd, feats, prefix="decoder.", num_queries=300, num_layers=6):
    """RTDETRTransformerv2.forward, eval path (rtdetrv2_decoder.py:782-814, :638-660, :695-780, :401-443)."""
    proj = [_cn(sd, f, f"{prefix}input_proj.{i}") for i, f in enumerate(feats)]
    shapes = [list(p.shape[2:]) for p in proj]
    memory = torch.concat([p.flatten(2).permute(0, 2, 1) for p in proj], 1)
    anchors, valid = sd[prefix + "anchors"], sd[prefix + "valid_mask"]
    om = _ln(sd, _lin(sd, valid.to(memory.dtype) * memory, prefix + "enc_output.proj"), prefix + "enc_output.norm")
    logits = _lin(sd, om, prefix + "enc_score_head")
    coords = _mlp(sd, om, prefix + "enc_bbox_head", 3) + anchors
    token_scores = logits.max(-1).values
    _, ind = torch.topk(token_scores, num_queries, dim=-1)
    # how far the last selected token's score lies above the first one left out: an implementation within 1e-3 of these logits
    # may legitimately select another token when this is smaller (tools/e2e_oracle_eval.py reads it)
    cut = torch.topk(token_scores, min(num_queries + 1, token_scores.shape[-1]), dim=-1).values
    topk_margin = (cut[:, num_queries - 1] - cut[:, -1]) if cut.shape[-1] > num_queries else torch.full((cut.shape[0],), float("inf"))
    target = om.gather(1, ind.unsqueeze(-1).repeat(1, 1, om.shape[-1]))
    ref_unact = coords.gather(1, ind.unsqueeze(-1).repeat(1, 1, 4))
    ref = torch.sigmoid(ref_unact)
    out = target
    for i in range(num_layers):
        p = f"{prefix}decoder.layers.{i}."
        qpe = _mlp(sd, ref, prefix + "query_pos_head", 2)
        q = out + qpe
        out = _ln(sd, out + _mha(sd, p + "self_attn.", 8, q, q, out), p + "norm1")
        out = _ln(sd, out + _deform_attn(sd, p + "cross_attn.", out + qpe, ref.unsqueeze(2), memory, shapes), p + "norm2")
        out = _ln(sd, out + _lin(sd, F.relu(_lin(sd, out, p + "linear1")), p + "linear2"), p + "norm3")
        box = torch.sigmoid(_mlp(sd, out, f"{prefix}dec_bbox_head.{i}", 3) + _inverse_sigmoid(ref))
        if i == num_layers - 1:
            return {"pred_logits": _lin(sd, out, f"{prefix}dec_score_head.{i}"), "pred_boxes": box, "topk_index": ind, "topk_margin": topk_margin}
        ref = box


@torch.inference_mode()
def rtdetr_forward(sd, x, num_queries=300):
    """models/rtdetr.py:16-21: fp32 N x 3 x S x S -> pred_logits N x num_queries x nc, pred_boxes N x num_queries x 4
    (S = 640 with 300 queries: layout / table structure; S = 960 with 1500: the cell detector; sd's anchors fix S)."""
    return rtdetr_decoder(sd, hybrid_encoder(sd, presnet(sd, x)), num_queries=num_queries)

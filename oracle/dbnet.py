"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference DBNet++ text detector as a function of a state dict, plain
PyTorch fp32.  Each function cites the reference lines it follows (paths relative to
/root/reference/src/yomitoku).

Third-party arithmetic not in the reference tree: torchvision 0.21.0 `models.resnet50`
(uv.lock:2595) - ResNet-50 v1.5, restated here from its published definition: stride on the 3x3
conv of each bottleneck, eval BatchNorm eps 1e-5, `replace_stride_with_dilation=[F,F,T]` turns
layer4's stride into dilation 2 for blocks >= 1 (block 0 keeps dilation 1, its downsample conv
gets stride 1).

Pinning: `oracle/pin_against_reference.py` runs the reference's own `DBNet` class (decoder and
ASF unmodified; torchvision replaced by this file's backbone because torchvision is not
installed) and checks equality with `dbnet_forward`; the vectors it writes live in
tests/golden/.  The backbone itself is therefore *parity unpinned* against torchvision.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

_LAYERS = ((64, 3), (128, 4), (256, 6), (512, 3))


def _bn(sd, x, name):
    # nn.BatchNorm2d in eval mode (torchvision resnet / dbnet_plus.py:112,115)
    return F.batch_norm(
        x, sd[name + ".running_mean"], sd[name + ".running_var"], sd[name + ".weight"], sd[name + ".bias"],
        False, 0.0, 1e-5,
    )


def _conv(sd, x, name, stride=1, padding=0, dilation=1):
    return F.conv2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride, padding, dilation)


def resnet50_dilated_features(sd, x, prefix="backbone.body."):
    """models/dbnet_plus.py:14-38: IntermediateLayerGetter over torchvision resnet50 with
    replace_stride_with_dilation=[False, False, True]; returns layer1..layer4 outputs."""
    x = F.relu(_bn(sd, _conv(sd, x, prefix + "conv1", 2, 3), prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    feats = []
    for li, (planes, blocks) in enumerate(_LAYERS, start=1):
        for bi in range(blocks):
            p = f"{prefix}layer{li}.{bi}."
            stride = 2 if (bi == 0 and li in (2, 3)) else 1
            dil = 2 if (li == 4 and bi > 0) else 1
            idn = x
            y = F.relu(_bn(sd, _conv(sd, x, p + "conv1"), p + "bn1"))
            y = F.relu(_bn(sd, _conv(sd, y, p + "conv2", stride, dil, dil), p + "bn2"))
            y = _bn(sd, _conv(sd, y, p + "conv3"), p + "bn3")
            if (p + "downsample.0.weight") in sd:
                idn = _bn(sd, _conv(sd, x, p + "downsample.0", stride), p + "downsample.1")
            x = F.relu(y + idn)
        feats.append(x)
    return feats


def asf(sd, concat_x, features, prefix="decoder.concat_attention."):
    """models/layers/dbnet_feature_attention.py:150-160 (ScaleFeatureSelection.forward) with
    attention_type="scale_channel_spatial" -> ScaleChannelSpatialAttention.forward :69-79."""
    x = _conv(sd, concat_x, prefix + "conv", 1, 1)
    ea = prefix + "enhanced_attention."
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(_conv(sd, g, ea + "channel_wise.1"))
    g = _conv(sd, g, ea + "channel_wise.3").sigmoid()
    gx = g + x
    m = torch.mean(gx, dim=1, keepdim=True)
    sp = _conv(sd, F.relu(_conv(sd, m, ea + "spatial_wise.0", 1, 1)), ea + "spatial_wise.2").sigmoid()
    gx = sp + gx
    score = _conv(sd, gx, ea + "attention_wise.0").sigmoid()
    return torch.cat([score[:, i : i + 1] * features[i] for i in range(len(features))], dim=1)


def dbnet_decoder(sd, feats, prefix="decoder."):
    """models/dbnet_plus.py:200-230 (DBNetDecoder.forward) + binarize head :110-119."""
    f = [_conv(sd, feats[i], f"{prefix}input_proj.layer{i + 1}") for i in range(4)]
    for i in (3, 2, 1):  # layer4 -> layer1 top-down
        bottom, top = f[i], f[i - 1]
        if bottom.shape[-2:] != top.shape[-2:]:
            bottom = F.interpolate(bottom, size=top.shape[-2:], mode="bilinear", align_corners=False)
        f[i - 1] = bottom + top
    fp = [_conv(sd, f[0], prefix + "out_proj.layer1", 1, 1)]
    for i, scale in ((1, 2), (2, 4), (3, 4)):
        y = _conv(sd, f[i], f"{prefix}out_proj.layer{i + 1}.0", 1, 1)
        fp.append(F.interpolate(y, scale_factor=scale, mode="bilinear", align_corners=False))
    rev = fp[::-1]
    fuse = asf(sd, torch.cat(rev, dim=1), rev, prefix + "concat_attention.")
    b = prefix + "binarize."
    y = F.relu(_bn(sd, _conv(sd, fuse, b + "0", 1, 1), b + "1"))
    y = F.conv_transpose2d(y, sd[b + "3.weight"], sd[b + "3.bias"], 2)
    y = F.relu(_bn(sd, y, b + "4"))
    y = F.conv_transpose2d(y, sd[b + "6.weight"], sd[b + "6.bias"], 2)
    return torch.sigmoid(y)


@torch.inference_mode()
def dbnet_forward(sd, x):
    """models/dbnet_plus.py:243-246: fp32 N x 3 x H x W -> {"binary": N x 1 x H x W}."""
    return {"binary": dbnet_decoder(sd, resnet50_dilated_features(sd, x))}

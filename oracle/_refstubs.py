"""ORACLE tooling (this container only): import the reference's Python modules from
/root/reference without their uninstalled third-party dependencies.

`yomitoku/__init__.py` pulls cv2, omegaconf, onnx, torchvision, timm ... none of which exist in
this image, so the sub-modules are loaded under namespace stubs that bypass the package
`__init__`s, and the missing third-party libraries are replaced by minimal restatements:

  torchvision.models.resnet50 / models._utils.IntermediateLayerGetter  (torchvision 0.21.0,
      uv.lock:2595) - ResNet-50 v1.5 as an nn.Module with torchvision's attribute names.
  timm (1.0.27) - see `_TimmViT` used for parseq pinning.
  omegaconf.ListConfig - a list subclass.

Nothing here is imported on the GPU box; /root/reference does not exist there.
"""

from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types
from collections import OrderedDict

import torch
import torch.nn as nn

REF_SRC = "/root/reference/src"


# ------------------------------------------------------------------ torchvision restatement
class _Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, dilation, dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idn = x
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        if self.downsample is not None:
            idn = self.downsample(x)
        return self.relu(out + idn)


class _ResNet50(nn.Module):
    def __init__(self, replace_stride_with_dilation=None, **_):
        super().__init__()
        rswd = replace_stride_with_dilation or [False, False, False]
        self.inplanes, self.dilation = 64, 1
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(64, 3)
        self.layer2 = self._make(128, 4, 2, rswd[0])
        self.layer3 = self._make(256, 6, 2, rswd[1])
        self.layer4 = self._make(512, 3, 2, rswd[2])
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, 1000)

    def _make(self, planes, blocks, stride=1, dilate=False):
        prev = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [_Bottleneck(self.inplanes, planes, stride, down, prev)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(_Bottleneck(self.inplanes, planes, dilation=self.dilation))
        return nn.Sequential(*layers)


class _IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        orig = dict(return_layers)
        remaining = dict(return_layers)
        layers = OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = orig

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out


# ------------------------------------------------------------------ timm restatement
class _PatchEmbed(nn.Module):
    def __init__(self, img_size, patch_size, in_chans, embed_dim):
        super().__init__()
        self.img_size = tuple(img_size)
        self.patch_size = tuple(patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size)
        self.norm = nn.Identity()

    def forward(self, x):
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))


class _TimmAttention(nn.Module):
    def __init__(self, dim, num_heads, qkv_bias):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x):
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, C // self.num_heads).permute(2, 0, 3, 1, 4)
        x = torch.nn.functional.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2])
        return self.proj(x.transpose(1, 2).reshape(B, N, C))


class _TimmMlp(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.act = nn.GELU()
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(self.act(self.fc1(x)))


class _TimmBlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio, qkv_bias):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=1e-6)
        self.attn = _TimmAttention(dim, num_heads, qkv_bias)
        self.norm2 = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _TimmMlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        return x + self.mlp(self.norm2(x))


class _TimmViT(nn.Module):
    """timm.models.vision_transformer.VisionTransformer, the subset the reference's Encoder uses
    (class_token=False, global_pool="", num_classes=0)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.0,
                 qkv_bias=True, drop_rate=0.0, attn_drop_rate=0.0, drop_path_rate=0.0, embed_layer=_PatchEmbed,
                 num_classes=0, global_pool="", class_token=False, **_):
        super().__init__()
        self.patch_embed = embed_layer(img_size, patch_size, in_chans, embed_dim)
        n = self.patch_embed.grid_size[0] * self.patch_embed.grid_size[1]
        self.pos_embed = nn.Parameter(torch.randn(1, n, embed_dim) * 0.02)
        self.pos_drop = nn.Identity()
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        self.blocks = nn.Sequential(*[_TimmBlock(embed_dim, num_heads, mlp_ratio, qkv_bias) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)

    def no_weight_decay(self):
        return {"pos_embed"}

    def forward_features(self, x):
        x = self.patch_embed(x)
        x = self.pos_drop(x + self.pos_embed)
        return self.norm(self.blocks(self.norm_pre(self.patch_drop(x))))


def _named_apply(fn, module, name="", depth_first=True, include_root=False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child in module.named_children():
        child_name = ".".join((name, child_name)) if name else child_name
        _named_apply(fn, child, child_name, depth_first, True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def _module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def install_stubs():
    """Register stand-ins for the third-party modules the reference imports but this image lacks."""
    if "torchvision" not in sys.modules:
        tv = _module("torchvision")
        tv.models = _module("torchvision.models", resnet50=lambda **kw: _ResNet50(**kw))
        tv.models._utils = _module("torchvision.models._utils", IntermediateLayerGetter=_IntermediateLayerGetter)
    if "cv2" not in sys.modules:  # imported at module level by reading_order.py / utils/misc.py, unused on the paths we call
        _module("cv2")
    if "timm" not in sys.modules:
        tm = _module("timm")
        tm.models = _module("timm.models")
        tm.models.vision_transformer = _module("timm.models.vision_transformer", PatchEmbed=_PatchEmbed,
                                               VisionTransformer=_TimmViT)
        tm.models.helpers = _module("timm.models.helpers", named_apply=_named_apply)
    if "omegaconf" not in sys.modules:

        class ListConfig(list):
            pass

        _module("omegaconf", ListConfig=ListConfig, OmegaConf=object)
    # namespace packages that skip yomitoku/__init__.py
    for pkg, sub in (
        ("yomitoku", "yomitoku"),
        ("yomitoku.models", "yomitoku/models"),
        ("yomitoku.models.layers", "yomitoku/models/layers"),
        ("yomitoku.postprocessor", "yomitoku/postprocessor"),
        ("yomitoku.utils", "yomitoku/utils"),
        ("yomitoku.data", "yomitoku/data"),
    ):
        if pkg not in sys.modules:
            m = types.ModuleType(pkg)
            m.__path__ = [os.path.join(REF_SRC, sub)]
            sys.modules[pkg] = m


def ref_import(modname: str):
    """Import a reference module (e.g. 'yomitoku.models.dbnet_plus') unmodified from /root/reference."""
    if not os.path.isdir(REF_SRC):
        raise RuntimeError("/root/reference is not available here")
    install_stubs()
    return importlib.import_module(modname)


class AttrDict(dict):
    """cfg stand-in: attribute access + ** expansion (what the reference uses OmegaConf for)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

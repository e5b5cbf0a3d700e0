"""Seeded synthetic checkpoints and inputs.

There is no network in the build/bench environment, so the pretrained safetensors of the
reference cannot be fetched.  These generators produce state dicts with exactly the reference's
parameter names and shapes, drawn from a seeded CPU generator so that the oracle,
the HIP path and the golden fixtures all see identical bytes.  The draws are variance preserving
(fan-in scaled convolutions, near-identity BatchNorm statistics, damped residual branches) so
that activations stay O(1) through 50 layers and the probability map / logits are not saturated -
otherwise a parity test would be vacuous.
"""

from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch


class _Draw:
    def __init__(self, seed: int):
        self.g = torch.Generator(device="cpu")
        self.g.manual_seed(int(seed))

    def normal(self, shape, std=1.0, mean=0.0):
        return torch.randn(*shape, generator=self.g, dtype=torch.float32) * std + mean

    def uniform(self, shape, lo, hi):
        return torch.rand(*shape, generator=self.g, dtype=torch.float32) * (hi - lo) + lo


def _conv(sd, d, name, cout, cin, k, bias=False, gain=2.0, kw=None):
    kw = k if kw is None else kw
    fan_in = cin * k * kw
    sd[name + ".weight"] = d.normal((cout, cin, k, kw), std=math.sqrt(gain / fan_in))
    if bias:
        sd[name + ".bias"] = d.normal((cout,), std=0.05)


def _bn(sd, d, name, c, gamma=(0.8, 1.2)):
    sd[name + ".weight"] = d.uniform((c,), *gamma)
    sd[name + ".bias"] = d.normal((c,), std=0.05)
    sd[name + ".running_mean"] = d.normal((c,), std=0.05)
    sd[name + ".running_var"] = d.uniform((c,), 0.8, 1.2)
    sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.int64)


def dbnet_state_dict(seed: int = 1234, hidden: int = 256, out_bias: float = 0.0) -> "OrderedDict[str, torch.Tensor]":
    """State dict of DBNet (reference models/dbnet_plus.py; torchvision resnet50 naming).
    `out_bias` shifts the last logit: a strongly negative value leaves only sparse blobs above the
    binarisation threshold (a realistic contour count for the post-processor under random weights)."""
    d = _Draw(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    bb = "backbone.body."
    _conv(sd, d, bb + "conv1", 64, 3, 7)
    _bn(sd, d, bb + "bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(zip((64, 128, 256, 512), (3, 4, 6, 3)), start=1):
        for bi in range(blocks):
            p = f"{bb}layer{li}.{bi}."
            _conv(sd, d, p + "conv1", planes, inplanes, 1)
            _bn(sd, d, p + "bn1", planes)
            _conv(sd, d, p + "conv2", planes, planes, 3)
            _bn(sd, d, p + "bn2", planes)
            _conv(sd, d, p + "conv3", planes * 4, planes, 1)
            _bn(sd, d, p + "bn3", planes * 4, gamma=(0.2, 0.4))  # damped residual branch
            if bi == 0:
                _conv(sd, d, p + "downsample.0", planes * 4, inplanes, 1, gain=1.0)
                _bn(sd, d, p + "downsample.1", planes * 4)
            inplanes = planes * 4
    dec = "decoder."
    q = hidden // 4
    for li, cin in enumerate((256, 512, 1024, 2048), start=1):
        _conv(sd, d, f"{dec}input_proj.layer{li}", hidden, cin, 1, gain=1.0)
    _conv(sd, d, dec + "out_proj.layer1", q, hidden, 3, gain=1.0)
    for li in (2, 3, 4):
        _conv(sd, d, f"{dec}out_proj.layer{li}.0", q, hidden, 3, gain=1.0)

    def head(prefix, cin):
        _conv(sd, d, prefix + ".0", q, cin, 3)
        _bn(sd, d, prefix + ".1", q)
        # ConvTranspose2d weights are [in][out][kh][kw]
        sd[prefix + ".3.weight"] = d.normal((q, q, 2, 2), std=math.sqrt(2.0 / q))
        sd[prefix + ".3.bias"] = d.normal((q,), std=0.05)
        _bn(sd, d, prefix + ".4", q)
        sd[prefix + ".6.weight"] = d.normal((q, 1, 2, 2), std=math.sqrt(1.0 / q) * 0.6)
        sd[prefix + ".6.bias"] = d.normal((1,), std=0.05) - 0.5

    head(dec + "binarize", hidden)
    sd[dec + "binarize.6.bias"] += out_bias
    # the `thresh` head exists in the checkpoint (adaptive=True, serial=True) but never runs in forward
    _conv(sd, d, dec + "thresh.0", q, hidden + 1, 3)
    _bn(sd, d, dec + "thresh.1", q)
    sd[dec + "thresh.3.weight"] = d.normal((q, q, 2, 2), std=math.sqrt(2.0 / q))
    sd[dec + "thresh.3.bias"] = d.normal((q,), std=0.05)
    _bn(sd, d, dec + "thresh.4", q)
    sd[dec + "thresh.6.weight"] = d.normal((q, 1, 2, 2), std=math.sqrt(1.0 / q))
    sd[dec + "thresh.6.bias"] = d.normal((1,), std=0.05)
    ca = dec + "concat_attention."
    _conv(sd, d, ca + "conv", q, hidden, 3, bias=True, gain=1.0)
    ea = ca + "enhanced_attention."
    _conv(sd, d, ea + "channel_wise.1", q // 4, q, 1)
    _conv(sd, d, ea + "channel_wise.3", q, q // 4, 1)
    _conv(sd, d, ea + "spatial_wise.0", 1, 1, 3)
    _conv(sd, d, ea + "spatial_wise.2", 1, 1, 1)
    _conv(sd, d, ea + "attention_wise.0", 4, q, 1, gain=1.0)
    return sd


# ---------------------------------------------------------------------------------------------
# synthetic inputs
def synthetic_page(seed: int = 0, height: int = 1600, width: int = 1200) -> np.ndarray:
    """uint8 BGR page: light background with dark text-like line blocks, a ruled table and a
    textured figure block.  Deterministic for a given (seed, height, width)."""
    rng = np.random.default_rng(seed)
    img = np.clip(rng.normal(245.0, 3.0, size=(height, width, 3)), 0, 255).astype(np.uint8)
    n_lines = int(rng.integers(40, 121))
    y = int(rng.integers(20, 60))
    for _ in range(n_lines):
        lh = int(rng.integers(16, 49))
        if y + lh + 10 >= height:
            break
        x0 = int(rng.integers(20, max(21, width // 6)))
        lw = int(rng.integers(80, min(1000, width - x0 - 20) + 1))
        x = x0
        while x < x0 + lw:  # glyph-like strokes
            gw = int(rng.integers(max(4, lh // 3), lh + 1))
            gx1 = min(x + gw, x0 + lw)
            ink = int(rng.integers(10, 70))
            block = img[y : y + lh, x:gx1]
            stroke = rng.random((lh, gx1 - x)) < 0.55
            block[stroke] = ink
            x = gx1 + int(rng.integers(2, max(3, lh // 4)))
        y += lh + int(rng.integers(6, 30))
    for _ in range(int(rng.integers(0, 3))):  # ruled tables
        th, tw = int(rng.integers(120, 360)), int(rng.integers(300, width - 100))
        ty, tx = int(rng.integers(0, height - th)), int(rng.integers(0, width - tw))
        rows, cols = int(rng.integers(2, 7)), int(rng.integers(2, 6))
        img[ty : ty + th, tx : tx + tw] = 250
        for r in range(rows + 1):
            yy = ty + (th - 2) * r // rows
            img[yy : yy + 2, tx : tx + tw] = 30
        for c in range(cols + 1):
            xx = tx + (tw - 2) * c // cols
            img[ty : ty + th, xx : xx + 2] = 30
    if rng.random() < 0.5:  # figure block
        fh, fw = int(rng.integers(150, 400)), int(rng.integers(200, 500))
        fy, fx = int(rng.integers(0, height - fh)), int(rng.integers(0, width - fw))
        img[fy : fy + fh, fx : fx + fw] = rng.integers(0, 256, size=(fh, fw, 3), dtype=np.uint8)
    return img


def synthetic_page_with_truth(seed: int = 0, height: int = 1600, width: int = 1200):
    """(page, line_quads, table_boxes, paragraph_boxes): a page laid out like a form - rows of text segments whose widths
    follow the crop-width distribution the reference quotes (log-normal, median ~120 px, cli/main.py:508),
    about a hundred segments per 1600x1200 page, plus 1-2 ruled tables.  The ground-truth quads let the
    benchmark drive the recogniser / table stages with a realistic unit count even though the seeded
    random detector / layout weights detect noise."""
    rng = np.random.default_rng(seed + 7919)
    img = np.clip(rng.normal(245.0, 3.0, size=(height, width, 3)), 0, 255).astype(np.uint8)
    lines, tables = [], []
    n_tables = int(rng.integers(1, 3))
    table_rows = []
    for _ in range(n_tables):
        th, tw = int(rng.integers(140, 300)), int(rng.integers(400, width - 120))
        ty, tx = int(rng.integers(60, height - th - 60)), int(rng.integers(20, width - tw - 20))
        if any(not (ty + th < a or ty > b) for a, b in table_rows):
            continue
        table_rows.append((ty - 10, ty + th + 10))
        rows, cols = int(rng.integers(3, 7)), int(rng.integers(2, 6))
        img[ty : ty + th, tx : tx + tw] = 250
        for r in range(rows + 1):
            yy = ty + (th - 2) * r // rows
            img[yy : yy + 2, tx : tx + tw] = 30
        for c in range(cols + 1):
            xx = tx + (tw - 2) * c // cols
            img[ty : ty + th, xx : xx + 2] = 30
        tables.append([tx, ty, tx + tw, ty + th])
        ch, cw = (th - 2) // rows, (tw - 2) // cols
        for r in range(rows):
            for c in range(cols):
                if rng.random() < 0.7 and ch >= 24 and cw >= 50:
                    lh = int(min(ch - 10, rng.integers(14, 27)))
                    lw = int(min(cw - 12, max(16, rng.lognormal(math.log(70), 0.5))))
                    x0, y0 = tx + c * cw + 6, ty + r * ch + 6
                    lines.append((x0, y0, lw, lh))
    y = int(rng.integers(20, 50))
    paragraphs = []  # blocks of consecutive text rows (what a layout model would call a paragraph)
    block, block_left = None, 0
    while y < height - 60:
        lh = int(rng.integers(16, 40))
        if any(a <= y + lh and y <= b for a, b in table_rows):
            y += lh + 8
            block = None
            continue
        if block is None or block_left == 0:
            block = [width, y, 0, y]
            paragraphs.append(block)
            block_left = int(rng.integers(2, 7))
        block_left -= 1
        x = int(rng.integers(20, 80))
        while x < width - 60:
            lw = int(np.clip(rng.lognormal(math.log(120), 0.8), 16, 800))
            if x + lw > width - 20:
                break
            lines.append((x, y, lw, lh))
            block[0], block[2], block[3] = min(block[0], x - 6), max(block[2], x + lw + 6), y + lh + 4
            x += lw + int(rng.integers(18, 120))
            if rng.random() < 0.35:
                break
        y += lh + int(rng.integers(6, 26))
    paragraphs = [[max(b[0], 0), max(b[1] - 4, 0), min(b[2], width), min(b[3], height)] for b in paragraphs if b[2] > b[0]]
    quads = []
    for x0, y0, lw, lh in lines:
        x = x0
        while x < x0 + lw:
            gw = int(rng.integers(max(4, lh // 3), lh + 1))
            gx1 = min(x + gw, x0 + lw)
            block = img[y0 : y0 + lh, x:gx1]
            block[rng.random((lh, gx1 - x)) < 0.55] = int(rng.integers(10, 70))
            x = gx1 + int(rng.integers(2, max(3, lh // 4)))
        quads.append([[x0 - 2, y0 - 2], [x0 + lw + 2, y0 - 2], [x0 + lw + 2, y0 + lh + 2], [x0 - 2, y0 + lh + 2]])
    return img, quads, tables, paragraphs


# ---------------------------------------------------------------------------------------------
def parseq_state_dict(seed: int = 1235, patch=(4, 8), enc_dim: int = 192, enc_depth: int = 12, enc_mlp: int = 4,
                      dec_dim: int = 192, dec_mlp: int = 4, num_tokens: int = 7121, max_label_length: int = 100,
                      img_size=(32, 800), eos_bias: float = 4.5, favour_token: int | None = None,
                      favour_bias: float = 0.0) -> "OrderedDict[str, torch.Tensor]":
    """State dict of PARSeq (reference models/parseq.py:61-78, parseq_transformer.py:43-57, timm ViT naming).

    `eos_bias` lifts the <eos> logit so that greedy decoding stops after a varying number of steps
    (random weights would otherwise almost never emit <eos>); `favour_token`/`favour_bias` lift one
    ordinary class to provoke the repetition early-stop (models/parseq.py:226-242)."""
    d = _Draw(seed)
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    ph, pw = patch
    gh, gw = img_size[0] // ph, img_size[1] // pw

    def lin(name, out, inp, gain=1.0, bias_std=0.02):
        sd[name + ".weight"] = d.normal((out, inp), std=gain / math.sqrt(inp))
        sd[name + ".bias"] = d.normal((out,), std=bias_std)

    def ln(name, c):
        sd[name + ".weight"] = d.uniform((c,), 0.8, 1.2)
        sd[name + ".bias"] = d.normal((c,), std=0.05)

    e = "encoder."
    sd[e + "pos_embed"] = d.normal((1, gh * gw, enc_dim), std=0.2)
    sd[e + "patch_embed.proj.weight"] = d.normal((enc_dim, 3, ph, pw), std=1.0 / math.sqrt(3 * ph * pw))
    sd[e + "patch_embed.proj.bias"] = d.normal((enc_dim,), std=0.02)
    for i in range(enc_depth):
        p = f"{e}blocks.{i}."
        ln(p + "norm1", enc_dim)
        lin(p + "attn.qkv", 3 * enc_dim, enc_dim, gain=1.5)
        lin(p + "attn.proj", enc_dim, enc_dim, gain=0.5)
        ln(p + "norm2", enc_dim)
        lin(p + "mlp.fc1", enc_mlp * enc_dim, enc_dim)
        lin(p + "mlp.fc2", enc_dim, enc_mlp * enc_dim, gain=0.5)
    ln(e + "norm", enc_dim)
    p = "decoder.layers.0."
    for att in ("self_attn", "cross_attn"):
        sd[p + att + ".in_proj_weight"] = d.normal((3 * dec_dim, dec_dim), std=1.5 / math.sqrt(dec_dim))
        sd[p + att + ".in_proj_bias"] = d.normal((3 * dec_dim,), std=0.02)
        lin(p + att + ".out_proj", dec_dim, dec_dim, gain=0.7)
    lin(p + "linear1", dec_mlp * dec_dim, dec_dim)
    lin(p + "linear2", dec_dim, dec_mlp * dec_dim, gain=0.5)
    for n in ("norm1", "norm2", "norm_q", "norm_c"):
        ln(p + n, dec_dim)
    ln("decoder.norm", dec_dim)
    lin("head", num_tokens - 2, dec_dim, gain=2.0, bias_std=0.1)
    sd["head.bias"][0] += eos_bias
    if favour_token is not None:
        sd["head.bias"][favour_token] += favour_bias
    sd["text_embed.embedding.weight"] = d.normal((num_tokens, dec_dim), std=1.0 / math.sqrt(dec_dim))
    sd["pos_queries"] = d.normal((1, max_label_length + 1, dec_dim), std=0.5)
    return sd


def synthetic_line_batch(seed: int, batch: int, width: int, height: int = 32) -> torch.Tensor:
    """fp32 B x 3 x H x W in [-1, 1] as ParseqDataset emits (ToTensor + Normalize(0.5, 0.5)):
    dark strokes on light paper up to a per-sample content width, then black (-1) canvas padding."""
    g = torch.Generator().manual_seed(seed)
    x = torch.full((batch, 3, height, width), -1.0)
    for b in range(batch):
        cw = int(torch.randint(max(8, width // 3), width + 1, (1,), generator=g))
        paper = 0.85 + 0.1 * torch.rand(1, height, cw, generator=g)
        ink = (torch.rand(1, height, cw, generator=g) < 0.3).float()
        img = paper * (1 - ink) + 0.1 * ink
        x[b, :, :, :cw] = (img.expand(3, -1, -1) + 0.02 * torch.randn(3, height, cw, generator=g)).clamp(0, 1) * 2 - 1
    return x


def synthetic_line_sheet(seed: int = 1, n_lines: int = 2048, sheet_width: int = 3200, line_height: int = 32):
    """(sheet, quads): `n_lines` text-line boxes, all `line_height` px tall, widths log-normal (median ~120 px, clipped to
    [16, 800]: the crop-width distribution the reference quotes, cli/main.py:508), packed row by row on one uint8 BGR
    sheet - BASELINE.json configs[2] ("2048 synthetic 32xW text-line crops") in a form TextRecognizer.__call__ accepts."""
    rng = np.random.default_rng(seed + 104729)
    widths = np.clip(rng.lognormal(math.log(120), 0.8, size=n_lines), 16, 800).astype(int)
    gap = 6
    rows, x, y = [], gap, gap
    for w in widths.tolist():
        if x + w + gap > sheet_width:
            x, y = gap, y + line_height + gap
        rows.append((x, y, w))
        x += w + gap
    height = y + line_height + gap
    img = np.clip(rng.normal(245.0, 3.0, size=(height, sheet_width, 3)), 0, 255).astype(np.uint8)
    quads = []
    for x0, y0, w in rows:
        xx = x0 + 2
        while xx < x0 + w - 2:  # glyph-like strokes
            gw = int(rng.integers(8, 25))
            x1 = min(xx + gw, x0 + w - 2)
            block = img[y0 + 3 : y0 + line_height - 3, xx:x1]
            block[rng.random(block.shape[:2]) < 0.55] = int(rng.integers(10, 70))
            xx = x1 + int(rng.integers(2, 7))
        quads.append([[x0, y0], [x0 + w, y0], [x0 + w, y0 + line_height], [x0, y0 + line_height]])
    return img, quads

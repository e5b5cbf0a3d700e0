"""Second-source checks of the oracle's restatements of libraries that are NOT installed here (cv2, pyclipper,
torchvision, timm), against independent implementations that ARE installed (torch, scipy, Pillow, transformers) or
against closed-form mathematics.  Each test says what it can and cannot prove; DESIGN.md section 5 has the table.

None of this touches the product path: it validates test infrastructure (oracle/), on the CPU.
"""
import math

import numpy as np
import pytest
import torch

from oracle import cvlike


# --------------------------------------------------------------------------------------------- INTER_AREA
def _area_by_integration(img, dw, dh):
    """cv2 INTER_AREA (shrinking) as mathematics: the image is a piecewise-constant function, a destination pixel is its
    mean over the source cell [d*s, min((d+1)*s, size)).  Separable integral via a prefix sum - no tap table."""
    def axis(a, dsize, ax):
        ssize = a.shape[ax]
        scale = ssize / dsize
        a = np.moveaxis(a.astype(np.float64), ax, 0)
        prefix = np.concatenate([np.zeros((1,) + a.shape[1:]), np.cumsum(a, axis=0)], 0)

        def F(t):  # integral of the step function over [0, t]
            i = min(int(math.floor(t)), ssize - 1)
            return prefix[i] + (t - i) * a[i]

        out = np.empty((dsize,) + a.shape[1:])
        for d in range(dsize):
            lo, hi = d * scale, min((d + 1) * scale, ssize)
            out[d] = (F(hi) - F(lo)) / (hi - lo)
        return np.moveaxis(out, 0, ax)

    return axis(axis(img, dw, 1), dh, 0)


@pytest.mark.parametrize("sh,sw,dh,dw", [(97, 131, 32, 41), (64, 200, 32, 77), (1200, 90, 1184, 90), (50, 50, 7, 49)])
def test_inter_area_fractional_vs_integration(sh, sw, dh, dw):
    """Proves: the coefficient tables (fractional edge weights, the clipped last cell) integrate the right cells.
    Cannot prove: cv2's float32 accumulation order (the restatement follows the published loop order)."""
    rng = np.random.default_rng(sh * 1000 + sw)
    img = rng.random((sh, sw, 3)).astype(np.float32) * 255
    got = cvlike.resize_area(img, (dw, dh))
    want = _area_by_integration(img, dw, dh)
    assert np.abs(got - want).max() < 2e-3  # float32 sums of up to ~40 x 255-valued terms
    u8 = rng.integers(0, 256, size=(sh, sw, 3), dtype=np.uint8)
    got8 = cvlike.resize_area(u8, (dw, dh)).astype(np.int64)
    want8 = _area_by_integration(u8, dw, dh)
    assert np.abs(got8 - want8).max() <= 0.5 + 1e-3  # rounded to the nearest integer (ties either way)


@pytest.mark.parametrize("k", [2, 3, 4])
def test_inter_area_integer_ratio_vs_torch_area(k):
    """Integer shrink factors: F.interpolate(mode='area') == adaptive average pooling == block means."""
    import torch.nn.functional as F

    rng = np.random.default_rng(k)
    img = rng.random((24 * k, 30 * k, 3)).astype(np.float32)
    got = cvlike.resize_area(img, (30, 24))
    want = F.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(24, 30), mode="area")[0].permute(1, 2, 0).numpy()
    assert np.abs(got - want).max() < 1e-5
    u8 = rng.integers(0, 256, size=(24 * k, 30 * k, 3), dtype=np.uint8)
    got8 = cvlike.resize_area(u8, (30, 24)).astype(np.float64)
    mean = u8.reshape(24, k, 30, k, 3).astype(np.float64).mean(axis=(1, 3))
    assert np.abs(got8 - mean).max() <= 0.5 + 1e-9
    if k == 2:  # the 2 x 2 fast path rounds half UP, (s + 2) >> 2, as the halving pyramid relies on
        assert np.array_equal(got8, np.floor(mean + 0.5))
        assert np.array_equal(cvlike.resize_half(u8), got8.astype(np.uint8))


def test_inter_area_vs_pillow_box_on_smooth_image():
    """Pillow's BOX filter is a different discretisation (whole source pixels by centre), so only closeness on a smooth
    image can be asserted - a sanity check that the geometry (which source span feeds which destination pixel) agrees."""
    from PIL import Image

    yy, xx = np.mgrid[0:300, 0:420]
    img = (127 + 100 * np.sin(xx / 40.0) * np.cos(yy / 55.0)).astype(np.uint8)
    got = cvlike.resize_area(np.repeat(img[:, :, None], 3, 2), (123, 77))[:, :, 0].astype(int)
    pil = np.asarray(Image.fromarray(img).resize((123, 77), Image.BOX)).astype(int)
    assert np.abs(got - pil).max() <= 3


# --------------------------------------------------------------------------------------------- findContours
def _random_blobs(seed, h=90, w=120):
    from scipy import ndimage

    rng = np.random.default_rng(seed)
    field = ndimage.gaussian_filter(rng.random((h, w)), 2.0)
    bitmap = field > np.quantile(field, 0.62)
    bitmap[rng.integers(0, h, 30), rng.integers(0, w, 30)] = True   # isolated pixels
    bitmap[rng.integers(0, h, 30), rng.integers(0, w, 30)] = False  # pin holes
    return bitmap


@pytest.mark.parametrize("seed", range(6))
def test_borders_vs_connected_components(seed):
    """Proves (Suzuki & Abe, 8-connected foreground): one outer border per 8-connected component, one hole border per
    4-connected background component that does not touch the frame, every chain walks 8-neighbour steps on foreground
    pixels, and the union of all chains is exactly the set of border points (foreground with a 4-neighbour background).
    Cannot prove: the ORDER cv2 returns RETR_LIST contours in (no library here fixes it) - order stays unpinned."""
    from scipy import ndimage

    bitmap = _random_blobs(seed)
    chains = cvlike.find_borders(bitmap)
    comp, n_comp = ndimage.label(bitmap, structure=np.ones((3, 3), int))
    bg, n_bg = ndimage.label(~np.pad(bitmap, 1), structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
    n_holes = n_bg - 1  # the padded frame's background component is not a hole
    assert len(chains) == n_comp + n_holes
    union = np.zeros_like(bitmap)
    owners = []
    for chain in chains:
        xs, ys = np.array([p[0] for p in chain]), np.array([p[1] for p in chain])
        assert bitmap[ys, xs].all()
        if len(chain) > 1:
            steps = np.abs(np.diff(np.array(chain + chain[:1]), axis=0)).max(axis=1)
            assert steps.max() == 1  # closed chain of 8-neighbour moves
        labels = set(comp[ys, xs].tolist())
        assert len(labels) == 1  # a border never leaves its component
        owners.append(labels.pop())
        union[ys, xs] = True
    assert set(owners) == set(range(1, n_comp + 1))
    four = ndimage.binary_erosion(bitmap, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]], border_value=0)
    assert np.array_equal(union, bitmap & ~four)
    # outer borders enclose their component: the bounding box of the component's first-found border is the component's
    for lab in range(1, n_comp + 1):
        ys, xs = np.nonzero(comp == lab)
        boxes = [(min(p[0] for p in c), min(p[1] for p in c), max(p[0] for p in c), max(p[1] for p in c))
                 for c, o in zip(chains, owners) if o == lab]
        assert (xs.min(), ys.min(), xs.max(), ys.max()) in boxes


# --------------------------------------------------------------------------------------------- minAreaRect
def _min_rect_bruteforce(points):
    """Minimum-area enclosing rectangle by brute force over the edges of scipy's (Qhull) convex hull."""
    from scipy.spatial import ConvexHull

    pts = np.unique(np.asarray(points, dtype=np.float64), axis=0)
    hull = pts[ConvexHull(pts).vertices]
    best = None
    for i in range(len(hull)):
        d = hull[(i + 1) % len(hull)] - hull[i]
        d /= np.hypot(*d)
        u, v = hull @ d, hull @ np.array([-d[1], d[0]])
        w, h = u.max() - u.min(), v.max() - v.min()
        if best is None or w * h < best[0] - 1e-12:
            best = (w * h, min(w, h), max(w, h))
    return best


@pytest.mark.parametrize("seed", range(8))
def test_min_area_rect_vs_qhull_bruteforce(seed):
    """Proves: area and side lengths of the rectangle (what `min_size` filters and `unclip` use).  Corners are compared
    as a point set only when the optimum is unique; cv2 returns float32 and so does the restatement."""
    rng = np.random.default_rng(seed)
    ang = rng.uniform(0, math.pi)
    base = rng.integers(-40, 40, size=(60, 2)) * [3, 1]
    rot = np.array([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]])
    pts = np.rint(base @ rot.T).astype(int) + 200
    corners, short = cvlike.min_area_rect([tuple(p) for p in pts])
    area, bshort, blong = _min_rect_bruteforce(pts)
    e1, e2 = np.hypot(*(corners[1] - corners[0])), np.hypot(*(corners[2] - corners[1]))
    assert abs(e1 * e2 - area) < 1e-2 * max(1.0, area) * 1e-2
    assert abs(short - bshort) < 1e-3 and abs(max(e1, e2) - blong) < 1e-3
    # every input point lies inside the returned rectangle (within float32 rounding)
    c = corners.astype(np.float64)
    ex, ey = (c[1] - c[0]) / max(e1, 1e-12), (c[2] - c[1]) / max(e2, 1e-12)
    rel = pts - c[0]
    assert (rel @ ex).min() > -1e-2 and (rel @ ex).max() < e1 + 1e-2
    assert (rel @ ey).min() > -1e-2 and (rel @ ey).max() < e2 + 1e-2


def test_min_area_rect_degenerate_inputs():
    c, s = cvlike.min_area_rect([(5, 7)])
    assert s == 0.0 and (c == [5, 7]).all()
    c, s = cvlike.min_area_rect([(0, 0), (10, 0), (4, 0)])
    assert s == 0.0


# --------------------------------------------------------------------------------------------- fillPoly + mean
@pytest.mark.parametrize("seed", range(5))
def test_polygon_mean_vs_pillow_rasteriser(seed):
    """cv2.fillPoly(mask, [contour]) + cv2.mean(pred, mask): the mask is the contour pixels plus the interior.  Pillow's
    polygon rasteriser (fill + outline) is an independent implementation of the same region for border-following chains
    (consecutive points are 8-neighbours, so the outline IS the chain)."""
    from PIL import Image, ImageDraw

    bitmap = _random_blobs(seed + 50)
    rng = np.random.default_rng(seed)
    pred = rng.random(bitmap.shape).astype(np.float32)
    chains = [c for c in cvlike.find_borders(bitmap) if len(c) >= 8]
    assert chains
    checked = 0
    for chain in chains[:12]:
        im = Image.new("L", (bitmap.shape[1], bitmap.shape[0]), 0)
        ImageDraw.Draw(im).polygon([tuple(p) for p in chain], fill=1, outline=1)
        mask = np.asarray(im).astype(bool)
        want = float(pred[mask].astype(np.float64).mean())
        got = cvlike.polygon_mean(pred, chain)
        # rasterisers may disagree on a few pixels at self-touching corners: the mean moves by at most that fraction
        assert abs(got - want) <= 3.0 / mask.sum() + 1e-9, (got, want, mask.sum())
        checked += 1
    assert checked >= 3


# --------------------------------------------------------------------------------------------- Clipper round offset
@pytest.mark.parametrize("w,h,delta,ang", [(200, 40, 12.5, 0.0), (90, 30, 7.3, 0.4), (300, 22, 18.9, 1.1), (50, 50, 3.2, 0.0)])
def test_round_offset_vs_minkowski_sum(w, h, delta, ang):
    """JT_ROUND offset of a rectangle = Minkowski sum with a disc of radius delta: every output vertex lies at distance
    delta from the rectangle (within Clipper's arc tolerance 0.25 and the integer rounding of vertices, 0.71), the
    enclosed area is A + P*delta + pi*delta^2 up to the polygonal approximation of the four quarter arcs, and its
    minimum-area rectangle is the rectangle grown by delta on each side (what the post-processor keeps).
    Cannot prove: Clipper's exact vertex count / rounding of individual vertices."""
    rot = np.array([[math.cos(ang), -math.sin(ang)], [math.sin(ang), math.cos(ang)]])
    rect = np.rint(np.array([[0, 0], [w, 0], [w, h], [0, h]]) @ rot.T + 300).astype(int)
    out = np.array(cvlike.offset_round(rect.tolist(), delta), dtype=np.float64)
    assert len(out) >= 8

    def dist_to_rect(p):
        best = 1e9
        for i in range(4):
            a, b = rect[i].astype(float), rect[(i + 1) % 4].astype(float)
            t = np.clip(np.dot(p - a, b - a) / np.dot(b - a, b - a), 0, 1)
            best = min(best, float(np.hypot(*(p - (a + t * (b - a))))))
        return best

    d = np.array([dist_to_rect(p) for p in out])
    assert d.min() > delta - 0.25 - 0.75 and d.max() < delta + 0.75
    x, y = out[:, 0], out[:, 1]
    area = 0.5 * abs(np.dot(x, np.roll(y, -1)) - np.dot(y, np.roll(x, -1)))
    e1, e2 = np.hypot(*(rect[1] - rect[0])), np.hypot(*(rect[2] - rect[1]))
    exact = e1 * e2 + 2 * (e1 + e2) * delta + math.pi * delta * delta
    # vertices are rounded to integers (each moves by <= 0.71 px): the area moves by at most ~half a pixel per unit of boundary
    assert abs(area - exact) < 0.5 * (2 * (e1 + e2) + 2 * math.pi * delta)
    corners, short = cvlike.min_area_rect([tuple(p) for p in out.astype(int)])
    sides = sorted([np.hypot(*(corners[1] - corners[0])), np.hypot(*(corners[2] - corners[1]))])
    assert abs(sides[0] - (min(e1, e2) + 2 * delta)) < 1.6 and abs(sides[1] - (max(e1, e2) + 2 * delta)) < 1.6


# --------------------------------------------------------------------------------------------- warpPerspective
def test_warp_perspective_translation_is_a_copy_and_bilinear_elsewhere():
    """For the axis-aligned integer quads DBNet emits, M is a translation and the warp must copy pixels; for a rotated
    quad the fixed-point bilinear result must stay within 1 grey level of float bilinear sampling (torch grid_sample)."""
    import torch.nn.functional as F

    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, size=(80, 120, 3), dtype=np.uint8)
    M = cvlike.perspective_transform([[10, 5], [90, 5], [90, 45], [10, 45]], [[0, 0], [80, 0], [80, 40], [0, 40]])
    assert np.array_equal(cvlike.warp_perspective(img, M, (80, 40)), img[5:45, 10:90])
    src = [[20, 10], [100, 22], [95, 60], [14, 50]]
    M = cvlike.perspective_transform(src, [[0, 0], [64, 0], [64, 32], [0, 32]])
    got = cvlike.warp_perspective(img, M, (64, 32)).astype(np.float64)
    Mi = np.linalg.inv(M)
    ys, xs = np.mgrid[0:32, 0:64]
    den = Mi[2, 0] * xs + Mi[2, 1] * ys + Mi[2, 2]
    sx, sy = (Mi[0, 0] * xs + Mi[0, 1] * ys + Mi[0, 2]) / den, (Mi[1, 0] * xs + Mi[1, 1] * ys + Mi[1, 2]) / den
    grid = torch.from_numpy(np.stack([(2 * sx + 1) / 120 - 1, (2 * sy + 1) / 80 - 1], -1)[None]).float()
    ref = F.grid_sample(torch.from_numpy(img).permute(2, 0, 1)[None].float(), grid, mode="bilinear", padding_mode="zeros",
                        align_corners=False)[0].permute(1, 2, 0).numpy()
    inside = (sx > 1) & (sx < 118) & (sy > 1) & (sy < 78)
    # 1/32-pixel source quantisation: the sample point moves by <= 1/64 px, i.e. <= ~4 grey levels on random noise
    assert np.abs(got - ref)[inside].max() <= 6.0
    assert np.abs(got - ref)[inside].mean() < 1.2


# --------------------------------------------------------------------------------------------- ResNet-50 v1.5 / ViT
def test_resnet50_restatement_matches_published_architecture():
    """torchvision.models.resnet50: 25 557 032 parameters, v1.5 (stride on the 3 x 3), state-dict names and shapes."""
    from oracle._refstubs import _ResNet50

    net = _ResNet50()
    assert sum(p.numel() for p in net.parameters()) == 25_557_032
    sd = net.state_dict()
    assert sd["conv1.weight"].shape == (64, 3, 7, 7) and sd["layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["layer2.0.conv2.weight"].shape == (128, 128, 3, 3) and sd["layer4.2.conv3.weight"].shape == (2048, 512, 1, 1)
    assert net.layer2[0].conv2.stride == (2, 2) and net.layer2[0].conv1.stride == (1, 1)  # v1.5
    d = _ResNet50(replace_stride_with_dilation=[False, False, True])
    assert d.layer4[0].conv2.stride == (1, 1) and d.layer4[0].conv2.dilation == (1, 1) and d.layer4[0].conv2.padding == (1, 1)
    assert d.layer4[1].conv2.dilation == (2, 2) and d.layer4[1].conv2.padding == (2, 2) and d.layer4[0].downsample[0].stride == (1, 1)
    y = d.layer4(torch.zeros(1, 1024, 10, 12))
    assert y.shape == (1, 2048, 10, 12)  # stride replaced by dilation: resolution kept


def test_resnet50_restatement_vs_transformers_resnet():
    """Hugging Face transformers ships an independent ResNet (v1.5 bottlenecks by default).  Same weights -> same
    features proves block wiring, stride placement, BN/ReLU order, the stem and the max-pool padding.  The dilated
    layer4 variant is not available there (checked structurally above)."""
    tr = pytest.importorskip("transformers")
    from oracle._refstubs import _ResNet50

    torch.manual_seed(0)
    net = _ResNet50().eval()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.1)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                          layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False)
    hf = tr.ResNetModel(cfg).eval()
    src = net.state_dict()
    dst = {}

    def put(d_prefix, s_conv, s_bn):
        dst[d_prefix + ".convolution.weight"] = src[s_conv + ".weight"]
        for k in ("weight", "bias", "running_mean", "running_var", "num_batches_tracked"):
            dst[d_prefix + ".normalization." + k] = src[s_bn + "." + k]

    put("embedder.embedder", "conv1", "bn1")
    for s, depth in enumerate([3, 4, 6, 3]):
        for b in range(depth):
            p, q = f"encoder.stages.{s}.layers.{b}", f"layer{s + 1}.{b}"
            for j in range(3):
                put(f"{p}.layer.{j}", f"{q}.conv{j + 1}", f"{q}.bn{j + 1}")
            if b == 0:
                put(f"{p}.shortcut", f"{q}.downsample.0", f"{q}.downsample.1")
    missing, unexpected = hf.load_state_dict(dst, strict=False)
    assert not unexpected and not [m for m in missing if "num_batches" not in m], (missing, unexpected)
    x = torch.randn(1, 3, 96, 128)
    with torch.no_grad():
        feats = hf(x, output_hidden_states=True).hidden_states
        t = net.maxpool(net.relu(net.bn1(net.conv1(x))))
        mine = []
        for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
            t = layer(t)
            mine.append(t)
    for a, b in zip(mine, feats[1:]):
        assert a.shape == b.shape and (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


def test_vit_restatement_parameter_count_and_block_vs_transformers():
    """timm VisionTransformer(class_token=False, num_classes=0, global_pool=''): parameter count from the published
    architecture, and one pre-LN block (LayerNorm eps 1e-6, fused qkv, exact GELU) against transformers' ViTLayer."""
    tr = pytest.importorskip("transformers")
    from oracle._refstubs import _TimmViT

    D, depth, heads, ph, pw = 192, 3, 6, 4, 8
    vit = _TimmViT(img_size=[32, 800], patch_size=[ph, pw], embed_dim=D, depth=depth, num_heads=heads, mlp_ratio=4)
    n_tok = (32 // ph) * (800 // pw)
    per_block = 2 * 2 * D + (3 * D * D + 3 * D) + (D * D + D) + (4 * D * D + 4 * D) + (4 * D * D + D)
    assert sum(p.numel() for p in vit.parameters()) == 3 * ph * pw * D + D + n_tok * D + depth * per_block + 2 * D
    cfg = tr.ViTConfig(hidden_size=D, num_attention_heads=heads, intermediate_size=4 * D, hidden_act="gelu", layer_norm_eps=1e-6,
                       qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    from transformers.models.vit.modeling_vit import ViTLayer

    layer = ViTLayer(cfg).eval()
    blk = vit.blocks[0].eval()
    torch.manual_seed(1)
    for p in blk.parameters():
        p.data.normal_(0, 0.2)
    sd = blk.state_dict()
    q, k, v = sd["attn.qkv.weight"].chunk(3, 0)
    qb, kb, vb = sd["attn.qkv.bias"].chunk(3, 0)
    names = set(layer.state_dict())
    if "attention.q_proj.weight" in names:  # transformers >= 5 naming
        a = {"q": "attention.q_proj", "k": "attention.k_proj", "v": "attention.v_proj", "o": "attention.o_proj",
             "fc1": "mlp.fc1", "fc2": "mlp.fc2"}
    else:  # transformers 4.x naming
        a = {"q": "attention.attention.query", "k": "attention.attention.key", "v": "attention.attention.value",
             "o": "attention.output.dense", "fc1": "intermediate.dense", "fc2": "output.dense"}
    mapped = {
        "layernorm_before.weight": sd["norm1.weight"], "layernorm_before.bias": sd["norm1.bias"],
        "layernorm_after.weight": sd["norm2.weight"], "layernorm_after.bias": sd["norm2.bias"],
        a["q"] + ".weight": q, a["q"] + ".bias": qb, a["k"] + ".weight": k, a["k"] + ".bias": kb,
        a["v"] + ".weight": v, a["v"] + ".bias": vb,
        a["o"] + ".weight": sd["attn.proj.weight"], a["o"] + ".bias": sd["attn.proj.bias"],
        a["fc1"] + ".weight": sd["mlp.fc1.weight"], a["fc1"] + ".bias": sd["mlp.fc1.bias"],
        a["fc2"] + ".weight": sd["mlp.fc2.weight"], a["fc2"] + ".bias": sd["mlp.fc2.bias"],
    }
    missing, unexpected = layer.load_state_dict(mapped, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    x = torch.randn(2, 37, D)
    with torch.no_grad():
        out = layer(x)
        out = out[0] if isinstance(out, tuple) else out
        ref = blk(x)
    assert (out - ref).abs().max().item() < 2e-4 * max(1.0, ref.abs().max().item())


def _hf_vit_layer(tr, D, heads, blk_sd):
    """A transformers ViTLayer (pre-LN, eps 1e-6, exact GELU, separate q / k / v) carrying one timm-named block."""
    from transformers.models.vit.modeling_vit import ViTLayer

    cfg = tr.ViTConfig(hidden_size=D, num_attention_heads=heads, intermediate_size=4 * D, hidden_act="gelu", layer_norm_eps=1e-6,
                       qkv_bias=True, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    try:
        cfg._attn_implementation = "eager"
    except Exception:
        pass
    layer = ViTLayer(cfg).eval()
    q, k, v = blk_sd["attn.qkv.weight"].chunk(3, 0)
    qb, kb, vb = blk_sd["attn.qkv.bias"].chunk(3, 0)
    if "attention.q_proj.weight" in set(layer.state_dict()):  # transformers >= 5 naming
        a = {"q": "attention.q_proj", "k": "attention.k_proj", "v": "attention.v_proj", "o": "attention.o_proj", "fc1": "mlp.fc1", "fc2": "mlp.fc2"}
    else:
        a = {"q": "attention.attention.query", "k": "attention.attention.key", "v": "attention.attention.value",
             "o": "attention.output.dense", "fc1": "intermediate.dense", "fc2": "output.dense"}
    mapped = {"layernorm_before.weight": blk_sd["norm1.weight"], "layernorm_before.bias": blk_sd["norm1.bias"],
              "layernorm_after.weight": blk_sd["norm2.weight"], "layernorm_after.bias": blk_sd["norm2.bias"],
              a["q"] + ".weight": q, a["q"] + ".bias": qb, a["k"] + ".weight": k, a["k"] + ".bias": kb, a["v"] + ".weight": v, a["v"] + ".bias": vb,
              a["o"] + ".weight": blk_sd["attn.proj.weight"], a["o"] + ".bias": blk_sd["attn.proj.bias"],
              a["fc1"] + ".weight": blk_sd["mlp.fc1.weight"], a["fc1"] + ".bias": blk_sd["mlp.fc1.bias"],
              a["fc2"] + ".weight": blk_sd["mlp.fc2.weight"], a["fc2"] + ".bias": blk_sd["mlp.fc2.bias"]}
    missing, unexpected = layer.load_state_dict(mapped, strict=False)
    assert not missing and not unexpected, (missing, unexpected)
    return layer


@pytest.mark.parametrize("geometry", [dict(D=192, heads=6, patch=(4, 8), depth=12, W=216),   # parseq-tiny-dynw-v4, a narrow mini-batch
                                      dict(D=192, heads=6, patch=(4, 8), depth=12, W=800),   # ... the full canvas
                                      dict(D=256, heads=4, patch=(8, 8), depth=3, W=344)])   # 8 x 8 patches (parseq / parseq-large)
def test_vit_encoder_whole_depth_and_dynamic_width_vs_transformers_chain(geometry):
    """oracle.parseq.vit_encode - the CPU reference of every PARSeq encoder test - against a chain that shares no code with
    it: patches cut with Tensor.unfold and projected by a plain matmul, the learned position table indexed token by token
    from (row, column) of the FULL grid (parseq_transformer.py:220-227: a narrow input keeps the columns it has), `depth`
    transformers ViTLayers, torch.nn.LayerNorm(eps 1e-6) at the end.  Proves the whole-depth wiring, the final norm and the
    cropped position rows - what the one-block check above cannot."""
    tr = pytest.importorskip("transformers")
    from oracle.parseq import make_cfg, vit_encode

    D, heads, (ph, pw), depth, W = geometry["D"], geometry["heads"], geometry["patch"], geometry["depth"], geometry["W"]
    g = torch.Generator().manual_seed(41)
    full_gh, full_gw = 32 // ph, 800 // pw
    sd = {"encoder.patch_embed.proj.weight": torch.randn(D, 3, ph, pw, generator=g) * 0.1, "encoder.patch_embed.proj.bias": torch.randn(D, generator=g) * 0.1,
          "encoder.pos_embed": torch.randn(1, full_gh * full_gw, D, generator=g) * 0.5,
          "encoder.norm.weight": torch.rand(D, generator=g) + 0.5, "encoder.norm.bias": torch.randn(D, generator=g) * 0.1}
    for i in range(depth):
        p = f"encoder.blocks.{i}."
        for name, shape, std in (("attn.qkv.weight", (3 * D, D), D ** -0.5), ("attn.proj.weight", (D, D), D ** -0.5),
                                 ("mlp.fc1.weight", (4 * D, D), D ** -0.5), ("mlp.fc2.weight", (D, 4 * D), (4 * D) ** -0.5)):
            sd[p + name] = torch.randn(shape, generator=g) * std
            sd[p + name.replace("weight", "bias")] = torch.randn(shape[0], generator=g) * 0.05
        for n in ("norm1", "norm2"):
            sd[p + n + ".weight"] = torch.rand(D, generator=g) + 0.5
            sd[p + n + ".bias"] = torch.randn(D, generator=g) * 0.1
    cfg = make_cfg(patch=(ph, pw), enc_dim=D, enc_heads=heads, enc_depth=depth)
    x = torch.randn(2, 3, 32, W, generator=g)
    with torch.no_grad():
        got = vit_encode(sd, cfg, x)
        gh, gw = 32 // ph, W // pw
        patches = x.unfold(2, ph, ph).unfold(3, pw, pw)                      # B, 3, gh, gw, ph, pw
        patches = patches.permute(0, 2, 3, 1, 4, 5).reshape(2, gh * gw, 3 * ph * pw)
        t = patches @ sd["encoder.patch_embed.proj.weight"].reshape(D, -1).T + sd["encoder.patch_embed.proj.bias"]
        rows = torch.tensor([r * full_gw + c for r in range(gh) for c in range(gw)])
        t = t + sd["encoder.pos_embed"][0, rows]
        for i in range(depth):
            p = f"encoder.blocks.{i}."
            out = _hf_vit_layer(tr, D, heads, {k[len(p):]: v for k, v in sd.items() if k.startswith(p)})(t)
            t = out[0] if isinstance(out, tuple) else out
        ln = torch.nn.LayerNorm(D, eps=1e-6)
        ln.weight.data, ln.bias.data = sd["encoder.norm.weight"], sd["encoder.norm.bias"]
        want = ln(t)
    assert got.shape == want.shape == (2, gh * gw, D)
    assert (got - want).abs().max().item() < 2e-4 * max(1.0, want.abs().max().item())


def test_dilated_layer4_vs_functional_composition_from_the_torchvision_spec():
    """torchvision.models.resnet50(replace_stride_with_dilation=[False, False, True]) (models/dbnet_plus.py:33-37), layer4
    as its _make_layer(dilate=True) builds it: the stride of 2 becomes a dilation - block 0 keeps dilation 1 (the
    `previous_dilation`) with stride 1 in its 3 x 3 AND in its 1 x 1 projection shortcut, blocks 1-2 run their 3 x 3 at
    dilation 2 / padding 2; Bottleneck v1.5 = conv1x1-BN-ReLU, conv3x3-BN-ReLU, conv1x1-BN, + identity, ReLU; eval BatchNorm
    eps 1e-5.  Composed here from F.conv2d and the BatchNorm formula only (no oracle/_refstubs, no oracle/dbnet code),
    started from layer3 features of transformers' independent ResNet, and compared with oracle.dbnet.resnet50_dilated_features."""
    tr = pytest.importorskip("transformers")
    from oracle._refstubs import _ResNet50
    from oracle.dbnet import resnet50_dilated_features

    g = torch.Generator().manual_seed(43)
    names = [(k, v.shape) for k, v in _ResNet50([False, False, True]).state_dict().items() if v.dtype.is_floating_point]
    sd = {}
    for k, shape in names:  # names and shapes are the published torchvision ones (checked against the parameter count above)
        if k.endswith("running_var"):
            sd[k] = torch.rand(shape, generator=g) + 0.5
        elif k.endswith("running_mean") or k.endswith("bias"):
            sd[k] = torch.randn(shape, generator=g) * 0.1
        elif len(shape) == 4:
            sd[k] = torch.randn(shape, generator=g) * (2.0 / (shape[1] * shape[2] * shape[3])) ** 0.5
        else:
            sd[k] = torch.rand(shape, generator=g) + 0.5
    x = torch.randn(1, 3, 96, 160, generator=g)
    with torch.no_grad():
        feats = resnet50_dilated_features({"backbone.body." + k: v for k, v in sd.items()}, x)

        # layers 1-3 from the independent implementation
        cfg = tr.ResNetConfig(num_channels=3, embedding_size=64, hidden_sizes=[256, 512, 1024, 2048], depths=[3, 4, 6, 3],
                              layer_type="bottleneck", hidden_act="relu", downsample_in_first_stage=False)
        hf = tr.ResNetModel(cfg).eval()
        dst = {}

        def put(d_prefix, s_conv, s_bn):
            dst[d_prefix + ".convolution.weight"] = sd[s_conv + ".weight"]
            for k in ("weight", "bias", "running_mean", "running_var"):
                dst[d_prefix + ".normalization." + k] = sd[s_bn + "." + k]

        put("embedder.embedder", "conv1", "bn1")
        for s, depth in enumerate([3, 4, 6, 3]):
            for b in range(depth):
                p, q = f"encoder.stages.{s}.layers.{b}", f"layer{s + 1}.{b}"
                for j in range(3):
                    put(f"{p}.layer.{j}", f"{q}.conv{j + 1}", f"{q}.bn{j + 1}")
                if b == 0:
                    put(f"{p}.shortcut", f"{q}.downsample.0", f"{q}.downsample.1")
        missing, unexpected = hf.load_state_dict(dst, strict=False)
        assert not unexpected and not [m for m in missing if "num_batches" not in m]
        hidden = hf(x, output_hidden_states=True).hidden_states
        assert (feats[2] - hidden[3]).abs().max().item() < 1e-4 * max(1.0, hidden[3].abs().max().item())

        def bn(t, p):
            shape = (1, -1, 1, 1)
            return (t - sd[p + ".running_mean"].view(shape)) / torch.sqrt(sd[p + ".running_var"].view(shape) + 1e-5) * sd[p + ".weight"].view(shape) \
                + sd[p + ".bias"].view(shape)

        t = hidden[3]  # layer3 output, 1024 channels at H/16
        for b, dil in enumerate((1, 2, 2)):
            p = f"layer4.{b}"
            y = torch.relu(bn(torch.nn.functional.conv2d(t, sd[p + ".conv1.weight"]), p + ".bn1"))
            y = torch.relu(bn(torch.nn.functional.conv2d(y, sd[p + ".conv2.weight"], stride=1, padding=dil, dilation=dil), p + ".bn2"))
            y = bn(torch.nn.functional.conv2d(y, sd[p + ".conv3.weight"]), p + ".bn3")
            idn = bn(torch.nn.functional.conv2d(t, sd[p + ".downsample.0.weight"], stride=1), p + ".downsample.1") if b == 0 else t
            t = torch.relu(y + idn)
    assert feats[3].shape == t.shape == (1, 2048, 6, 10)  # H/16: the stride was replaced, the resolution kept
    assert (feats[3] - t).abs().max().item() < 1e-4 * max(1.0, t.abs().max().item())

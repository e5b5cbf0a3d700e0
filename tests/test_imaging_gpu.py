"""Pre-processing kernels vs the oracle chains on the same pages.
  detector: fp32 within 2e-6 of the area-resize restatement (summation order only);
  RT-DETR inputs: bit-exact against Pillow itself (8-bit two-pass resample);
  recogniser crops: uint8-exact pixels (so the normalised fp32 tensors are equal)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _page(seed, h, w):
    from yomitoku_amd.utils.synth import synthetic_page

    return synthetic_page(seed, h, w)


@pytest.mark.parametrize("h,w", [(1600, 1200), (1200, 1600), (480, 640), (2100, 1500), (91, 38), (700, 2400)])
def test_detector_preprocess(dev, h, w):
    from oracle.preprocess import detector_preprocess
    from yomitoku_amd import imaging

    img = np.ascontiguousarray(_page(h + w, max(h, 600), max(w, 600))[:h, :w])
    ref = detector_preprocess(img)
    out = imaging.detector_tensor(imaging.page_to_device(img, dev), 1280, 1600).cpu()
    assert out.shape == ref.shape
    assert (out - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("h,w,box", [(1600, 1200, None), (1200, 1600, None), (1600, 1200, (100, 200, 900, 700)),
                                     (800, 600, (3, 5, 77, 41)), (640, 640, None), (2000, 1400, (0, 0, 1400, 2000))])
def test_rtdetr_preprocess_is_pillow_exact(dev, h, w, box):
    from oracle.preprocess import rtdetr_preprocess
    from yomitoku_amd import imaging

    img = _page(h, h, w)
    ref, size = rtdetr_preprocess(img, box)
    out, osize, _ = imaging.rtdetr_tensor(imaging.page_to_device(img, dev), box)
    assert osize == size
    assert torch.equal(out.cpu(), ref[0])


def test_rtdetr_batch_of_crops_is_pillow_exact(dev):
    """imaging.rtdetr_batch_tensor (every crop of a forward in ONE launch, one blob of coefficient tables): each crop equals
    Pillow's resize of it - crops of two pages of different sizes, whole pages, repeated sizes, out-of-range boxes clamped as
    numpy slices them, a second call through the other staging buffer."""
    from oracle.preprocess import rtdetr_preprocess
    from yomitoku_amd import imaging

    imgs = [_page(11, 1600, 1200), _page(12, 1000, 1400)]
    pages = [imaging.page_to_device(img, dev) for img in imgs]
    crops = [(0, None), (1, None), (0, (100, 200, 900, 700)), (1, (3, 5, 77, 41)), (1, (200, 100, 1400, 1000)), (0, (100, 200, 900, 700)),
             (0, (-5, -7, 300, 2000)), (1, (640, 0, 1280, 640))]
    for turn in range(10):  # ten calls: every pinned staging buffer of the thread's ring, and two of them a second time
        turn %= 5
        out, metas = imaging.rtdetr_batch_tensor(pages, crops[turn:])
        assert out.shape == (len(crops) - turn, 3, 640, 640)
        for k, (p, box) in enumerate(crops[turn:]):
            clamped = None if box is None else (max(box[0], 0), max(box[1], 0), min(box[2], imgs[p].shape[1]), min(box[3], imgs[p].shape[0]))
            ref, size = rtdetr_preprocess(imgs[p], clamped)
            assert metas[k]["size"] == size and metas[k]["offset"] == ((0, 0) if clamped is None else clamped[:2])
            assert torch.equal(out[k].cpu(), ref[0]), (turn, k)
            one, osize, off = imaging.rtdetr_tensor(pages[p], box)
            assert torch.equal(one, out[k]) and osize == size and off == metas[k]["offset"]
    empty, metas = imaging.rtdetr_batch_tensor(pages, [])
    assert empty.shape == (0, 3, 640, 640) and metas == []
    with pytest.raises(ValueError):
        imaging.rtdetr_batch_tensor(pages, [(0, (50, 50, 50, 90))])


def test_to_host_hands_out_copies_equal_to_cpu(dev):
    """imaging.to_host: several device tensors of different dtypes through one pinned buffer - equal to `.cpu()`, owned by the
    caller (a second call does not change the first call's arrays), odd sizes, a non-contiguous view."""
    from yomitoku_amd import imaging

    g = torch.Generator().manual_seed(4)
    a = torch.randint(0, 7000, (37, 101), generator=g, dtype=torch.int32).to(dev)
    b = torch.rand((37, 101), generator=g).to(dev)
    c = torch.rand((5, 3, 7), generator=g).to(dev).permute(2, 0, 1)
    ha, hb, hc = imaging.to_host(a, b, c)
    again = imaging.to_host(b * 2.0)[0]
    assert np.array_equal(ha, a.cpu().numpy()) and np.array_equal(hb, b.cpu().numpy()) and np.array_equal(hc, c.cpu().numpy())
    assert np.array_equal(again, (b * 2.0).cpu().numpy()) and ha.flags.owndata and hb.dtype == np.float32 and hc.shape == (7, 5, 3)
    assert imaging.to_host() == ()


def _quads(rng, h, w, n):
    quads = []
    for k in range(n):
        kind = k % 5
        x, y = int(rng.integers(5, w - 420)), int(rng.integers(5, h - 320))
        if kind == 0:  # upright line
            bw, bh = int(rng.integers(40, 400)), int(rng.integers(14, 31))
            quads.append([[x, y], [x + bw, y], [x + bw, y + bh], [x, y + bh]])
        elif kind == 1:  # taller than 32: down-scaled
            bw, bh = int(rng.integers(100, 400)), int(rng.integers(40, 120))
            quads.append([[x, y], [x + bw, y], [x + bw, y + bh], [x, y + bh]])
        elif kind == 2:  # vertical line: rotated by 90 degrees
            bw, bh = int(rng.integers(14, 40)), int(rng.integers(100, 300))
            quads.append([[x, y], [x + bw, y], [x + bw, y + bh], [x, y + bh]])
        elif kind == 3:  # slanted quad
            bw, bh, sk = int(rng.integers(80, 300)), int(rng.integers(16, 50)), int(rng.integers(-12, 13))
            quads.append([[x, y + 12], [x + bw, y + 12 + sk], [x + bw - 3, y + 12 + sk + bh], [x - 2, y + 12 + bh]])
        else:  # exact 2x / 3x integer factors
            f = int(rng.integers(2, 4))
            bw, bh = 64 * f, 32 * f
            quads.append([[x, y], [x + bw, y], [x + bw, y + bh], [x, y + bh]])
    return quads


@pytest.mark.parametrize("dynamic", [False, True])
def test_recogniser_crops_match_oracle(dev, dynamic):
    from oracle.preprocess import parseq_crop
    from yomitoku_amd import imaging

    rng = np.random.default_rng(5)
    img = _page(11, 1000, 1400)
    quads = _quads(rng, 1000, 1400, 25)
    page = imaging.page_to_device(img, dev)
    plans = imaging.plan_crops(img.shape[:2], quads, (32, 800), dynamic)
    assert all(p is not None for p in plans)
    rgb = img[:, :, ::-1]
    for start in range(0, len(plans), 7):
        chunk = plans[start : start + 7]
        batch = imaging.build_crop_batch(page, chunk, 32, None if dynamic else 800).cpu()
        for slot, plan in enumerate(chunk):
            ref, cw = parseq_crop(rgb, quads[plan.index], (32, 800), dynamic)
            assert cw == plan.content_width and ref.shape[-1] == plan.canvas_width
            got = batch[slot, :, :, : ref.shape[-1]]
            assert torch.equal(got, ref), f"crop {plan.index} differs: {(got - ref).abs().max().item()}"
            assert (batch[slot, :, :, ref.shape[-1]:] == -1).all()


def test_recogniser_crops_large_downscale(dev):
    """Crops scaled down by more than the 24-tap table of the detector path (a ~900 px tall line, a 2000 px tall
    vertical one, and the whole page as TextRecognizer(img, points=None) hands it over): uint8-exact vs the oracle."""
    from oracle.preprocess import parseq_crop
    from yomitoku_amd import imaging

    img = _page(12, 2100, 1500)
    h, w = img.shape[:2]
    quads = [[[10, 20], [1480, 20], [1480, 925], [10, 925]],      # 905 px tall -> scale 28.3 on both axes
             [[40, 30], [740, 30], [740, 2090], [40, 2090]],      # vertical: rotated, 2060 -> 800 wide, 700 -> 32 tall
             [[0, 0], [w, 0], [w, h], [0, h]]]                    # the whole page
    page = imaging.page_to_device(img, dev)
    rgb = img[:, :, ::-1]
    for dynamic in (False, True):
        plans = imaging.plan_crops(img.shape[:2], quads, (32, 800), dynamic)
        assert all(p is not None for p in plans)
        batch = imaging.build_crop_batch(page, plans, 32, None if dynamic else 800).cpu()
        for slot, plan in enumerate(plans):
            ref, cw = parseq_crop(rgb, quads[plan.index], (32, 800), dynamic)
            assert cw == plan.content_width
            got = batch[slot, :, :, : ref.shape[-1]]
            assert torch.equal(got, ref), f"crop {plan.index} differs: {(got - ref).abs().max().item()}"


def test_invalid_quads_are_dropped():
    from yomitoku_amd import imaging

    plans = imaging.plan_crops((100, 200), [[[0, 0], [10, 0], [10, 10]], [[0, 0], [300, 0], [300, 10], [0, 10]],
                                            [[5, 5], [50, 5], [50, 20], [5, 20]]], (32, 800), True)
    assert plans[0] is None and plans[1] is None and plans[2] is not None


@pytest.mark.parametrize("h,w", [(1000, 1400), (999, 1399), (501, 303), (4, 6)])
def test_pyramid_halving_matches_oracle(dev, h, w):
    """source_downscale pyramid step (data/dataset.py:73-79): fast 2x2 cells, partial cells on odd sizes."""
    from oracle.cvlike import resize_half
    from yomitoku_amd import imaging

    img = np.ascontiguousarray(_page(h * 3 + w, max(h, 600), max(w, 600))[:h, :w])
    levels = imaging.build_pyramid(imaging.page_to_device(img, dev), [2])
    assert len(levels) == 3 and np.array_equal(levels[1].cpu().numpy(), resize_half(img))
    want = resize_half(resize_half(img))
    assert levels[2].shape == want.shape
    assert np.array_equal(levels[2].cpu().numpy(), want)


def _big_quads(rng, h, w):
    """Text boxes whose short side lands on pyramid levels 0..3 (32 px canvas: 64 / 128 / 256 px thresholds)."""
    quads = []
    for short, long_ in ((24, 180), (40, 260), (70, 300), (100, 420), (140, 520), (200, 640), (270, 900), (300, 420)):
        for _ in range(3):
            x, y = int(rng.integers(4, w - long_ - 8)), int(rng.integers(4, h - short - 8))
            d = rng.integers(-3, 4, size=(4, 2))
            quads.append([[x + int(d[0, 0]), y + int(d[0, 1])], [x + long_ + int(d[1, 0]), y + int(d[1, 1])],
                          [x + long_ + int(d[2, 0]), y + short + int(d[2, 1])], [x + int(d[3, 0]), y + short + int(d[3, 1])]])
    quads.append([[30, 40], [130, 40], [130, 700], [30, 700]])  # vertical line: rotated crop from level 1
    return quads


@pytest.mark.parametrize("shape", [(1000, 1400), (999, 1399)])
def test_source_downscale_crops_match_oracle(dev, shape):
    from oracle import cvlike
    from oracle.preprocess import calc_source_levels, parseq_crop
    from yomitoku_amd import imaging

    h, w = shape
    img = np.ascontiguousarray(_page(23, 1000, 1400)[:h, :w])
    quads = _big_quads(np.random.default_rng(9), h, w)
    page = imaging.page_to_device(img, dev)
    plans, levels = imaging.plan_crops_pyramid(img.shape[:2], quads, (32, 800), True, source_downscale=True)
    assert levels.tolist() == calc_source_levels(quads, 32).tolist() and set(levels.tolist()) == {0, 1, 2, 3}
    pyramid = imaging.build_pyramid(page, levels)
    ref_levels = {0: img}
    for k in range(1, 4):
        ref_levels[k] = cvlike.resize_half(ref_levels[k - 1])
    assert all(p is not None for p in plans)
    for start in range(0, len(plans), 6):
        chunk = plans[start : start + 6]
        batch = imaging.build_crop_batch(pyramid, chunk, 32, None).cpu()
        for slot, plan in enumerate(chunk):
            k = int(levels[plan.index])
            assert plan.desc.level == k
            q = quads[plan.index] if k == 0 else (np.asarray(quads[plan.index], dtype=np.float32) / (2.0 ** k)).tolist()
            ref, cw = parseq_crop(ref_levels[k][:, :, ::-1], q, (32, 800), True)
            assert cw == plan.content_width and ref.shape[-1] == plan.canvas_width
            got = batch[slot, :, :, : ref.shape[-1]]
            assert torch.equal(got, ref), f"crop {plan.index} (level {k}) differs: {(got - ref).abs().max().item()}"


def test_flipped_crops_match_oracle(dev):
    """Orientation-fallback retry input: the (rotated) ROI turned by 180 degrees, fixed 800 px canvas."""
    from oracle.preprocess import canvas_tensor, parseq_crop
    from yomitoku_amd import imaging

    rng = np.random.default_rng(6)
    img = _page(12, 1000, 1400)
    quads = _quads(rng, 1000, 1400, 14) + [[[30, 40], [70, 40], [70, 400], [30, 400]]]
    page = imaging.page_to_device(img, dev)
    plans = imaging.plan_crops(img.shape[:2], quads, (32, 800), True)
    batch = imaging.build_crop_batch(page, plans, 32, 800, flip=True).cpu()
    rgb = img[:, :, ::-1]
    for slot, plan in enumerate(plans):
        _, _, roi = parseq_crop(rgb, quads[plan.index], (32, 800), True, with_roi=True)
        ref, _ = canvas_tensor(np.ascontiguousarray(np.rot90(roi, 2)), (32, 800))
        assert torch.equal(batch[slot], ref), f"flipped crop {plan.index} differs"

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


def test_invalid_quads_are_dropped():
    from yomitoku_amd import imaging

    plans = imaging.plan_crops((100, 200), [[[0, 0], [10, 0], [10, 10]], [[0, 0], [300, 0], [300, 10], [0, 10]],
                                            [[5, 5], [50, 5], [50, 20], [5, 20]]], (32, 800), True)
    assert plans[0] is None and plans[1] is None and plans[2] is not None

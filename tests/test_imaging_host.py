"""Host half of the recogniser crop path (no GPU): the page-at-once crop planner must produce the same
descriptors, byte for byte, as the per-quad form that follows the reference's helpers statement by statement
(data/functions.py extract_roi_with_perspective / rotate_text_image / resize_with_padding)."""
import numpy as np
import pytest

from yomitoku_amd import imaging


def _quads(seed, n=200):
    rng = np.random.default_rng(seed)
    qs = []
    for _ in range(n):
        x, y = int(rng.integers(8, 1400)), int(rng.integers(8, 1000))
        w, h = int(rng.integers(5, 400)), int(rng.integers(5, 200))
        if rng.random() < 0.2:
            w, h = h // 3 + 4, w  # vertical lines (rotated crops)
        d = rng.integers(-6, 7, size=(4, 2)).tolist()
        qs.append([[x + d[0][0], y + d[0][1]], [x + w + d[1][0], y + d[1][1]], [x + w + d[2][0], y + h + d[2][1]],
                   [x + d[3][0], y + h + d[3][1]]])
    qs.append([[0, 0], [10, 0], [10, 10]])                                 # not a quad
    qs.append([[1.7, 2.2], [50.9, 2.1], [50.2, 20.8], [1.1, 20.9]])        # float corners truncate
    qs.append([[1500, 1100], [1700, 1100], [1700, 1150], [1500, 1150]])    # leaves the page
    qs.append([[10, 10], [1400, 10], [1400, 40], [10, 40]])                # wider than the 800 px canvas
    return qs


@pytest.mark.parametrize("dynamic", [False, True])
@pytest.mark.parametrize("seed", [0, 1])
def test_batched_planner_equals_per_quad_form(seed, dynamic):
    qs = _quads(seed)
    fast = imaging.plan_crops((1200, 1600), qs, (32, 800), dynamic)
    slow = imaging._plan_crops_scalar((1200, 1600), qs, (32, 800), dynamic)
    assert len(fast) == len(slow) == len(qs)
    live = 0
    for a, b in zip(fast, slow):
        assert (a is None) == (b is None)
        if a is None:
            continue
        live += 1
        assert (a.index, a.content_width, a.canvas_width) == (b.index, b.content_width, b.canvas_width)
        assert bytes(a.desc) == bytes(b.desc), (a.index, list(a.desc.minv), list(b.desc.minv))
    assert live >= 190


def test_planner_edge_cases():
    assert imaging.plan_crops((100, 100), []) == []
    assert imaging.plan_crops((100, 100), [[[0, 0], [1, 1]]]) == [None]
    # a zero-width / zero-height quad is dropped like an invalid one (it does not abort the page); its neighbours survive
    quads = [[[5, 5], [5, 5], [5, 9], [5, 9]], [[5, 5], [60, 5], [60, 25], [5, 25]], [[10, 10], [60, 10], [60, 10], [10, 10]]]
    for planner in (imaging.plan_crops, imaging._plan_crops_scalar):
        plans = planner((100, 100), quads)
        assert [p is None for p in plans] == [True, False, True] and plans[1].index == 1
    assert imaging.plan_crops((100, 100), quads[:1]) == [None]


def test_pil_coefficient_tables_in_array_form_equal_the_loop():
    """imaging.pil_bilinear_coeffs (all outputs at once, float64, weights summed tap by tap) against the statement-for-statement
    form of Pillow's precompute_coeffs / normalize_coeffs_8bpc it replaced: bounds and 22-bit coefficients equal, bit for bit,
    at up-scales, down-scales by 1 ... 8, odd sizes and the sizes around a power of two (a table crop has any size)."""
    import random

    from yomitoku_amd import imaging as im

    rng = random.Random(1)
    cases = [(1, 640), (2, 640), (37, 640), (333, 640), (639, 640), (640, 640), (641, 640), (1000, 640), (1279, 640), (1280, 640), (1281, 640),
             (1400, 640), (4800, 640), (5000, 640), (100, 960), (2000, 960), (7, 3), (3, 7)] + [(rng.randint(1, 3000), 640) for _ in range(60)]
    for a, b in cases:
        want, got = im._pil_bilinear_coeffs_scalar(a, b), im.pil_bilinear_coeffs(a, b)
        assert want[2] == got[2] and got[0].dtype == np.int32 and got[1].dtype == np.int32, (a, b)
        assert np.array_equal(want[0], got[0]) and np.array_equal(want[1], got[1]), (a, b)

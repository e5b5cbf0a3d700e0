"""CPU parity of the C++ DB post-processor (ymk_db_postprocess, host code of libymk_hip.so) against
the oracle restatement (oracle/cvlike.db_postprocess) on synthetic probability maps - blurred
upright and rotated text blobs, blobs with holes, specks, blobs touching the border.  Quads are
integers: they must match bit for bit (north_star "bit-exact box indices"); scores to 1e-9."""
import ctypes

import numpy as np
import pytest

from yomitoku_amd import _lib


def _blur(a, k=5):
    ker = np.ones(k, dtype=np.float32) / k
    a = np.apply_along_axis(lambda r: np.convolve(r, ker, mode="same"), 1, a)
    return np.apply_along_axis(lambda r: np.convolve(r, ker, mode="same"), 0, a).astype(np.float32)


def synthetic_prob_map(seed, h=320, w=416):
    rng = np.random.default_rng(seed)
    m = np.zeros((h, w), dtype=np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(int(rng.integers(6, 16))):
        cx, cy = rng.uniform(20, w - 20), rng.uniform(15, h - 15)
        bw, bh = rng.uniform(15, 140), rng.uniform(6, 26)
        ang = rng.choice([0.0, 0.0, rng.uniform(-0.5, 0.5), np.pi / 2])
        c, s = np.cos(ang), np.sin(ang)
        u = (xx - cx) * c + (yy - cy) * s
        v = -(xx - cx) * s + (yy - cy) * c
        m = np.maximum(m, ((np.abs(u) < bw / 2) & (np.abs(v) < bh / 2)).astype(np.float32) * rng.uniform(0.6, 1.0))
    if rng.random() < 0.7:  # a blob with a hole
        x0, y0 = int(rng.integers(10, w - 90)), int(rng.integers(10, h - 60))
        m[y0 : y0 + 40, x0 : x0 + 70] = 0.9
        m[y0 + 12 : y0 + 26, x0 + 20 : x0 + 50] = 0.0
    m[0:6, 30:90] = 0.95  # touches the top border
    m = _blur(m, 5)
    specks = rng.random((h, w)) > 0.9995
    m[specks] = 0.8
    return np.clip(m + rng.normal(0, 0.01, (h, w)).astype(np.float32), 0, 1).astype(np.float32)


def run_cpp(pred, image_size, min_size, thresh, box_thresh, max_candidates, unclip_ratio):
    lib = _lib.load()
    h, w = pred.shape
    cap = 4096
    quads = np.zeros((cap, 4, 2), dtype=np.int16)
    scores = np.zeros(cap, dtype=np.float64)
    n = ctypes.c_int()
    pred = np.ascontiguousarray(pred, dtype=np.float32)
    _lib.check(
        lib.ymk_db_postprocess(pred.ctypes.data, h, w, thresh, box_thresh, min_size, max_candidates, unclip_ratio,
                               int(image_size[1]), int(image_size[0]), quads.ctypes.data, scores.ctypes.data, cap,
                               ctypes.byref(n)),
        "ymk_db_postprocess",
    )
    return quads[: n.value].tolist(), scores[: n.value].tolist()


@pytest.mark.parametrize("seed", range(12))
@pytest.mark.parametrize("params", [(2, 0.3, 0.4, 1500, 3.5), (2, 0.15, 0.5, 1500, 7.0)])
def test_cpp_matches_oracle(seed, params):
    from oracle.cvlike import db_postprocess

    pred = synthetic_prob_map(seed)
    size = (1600, 1200) if seed % 2 else (int(pred.shape[0] * 1.7), int(pred.shape[1] * 2.3))
    ref_q, ref_s = db_postprocess(pred, size, *params)
    q, s = run_cpp(pred, size, *params)
    assert len(ref_q) >= 5
    assert q == ref_q
    assert np.allclose(s, ref_s, rtol=0, atol=1e-9)


def test_empty_and_full_maps():
    from oracle.cvlike import db_postprocess

    for fill in (0.0, 1.0):
        pred = np.full((64, 96), fill, dtype=np.float32)
        q, s = run_cpp(pred, (128, 192), 2, 0.3, 0.4, 1500, 3.5)
        rq, rs = db_postprocess(pred, (128, 192), 2, 0.3, 0.4, 1500, 3.5)
        assert q == rq and np.allclose(s, rs)
    assert run_cpp(np.zeros((64, 96), np.float32), (64, 96), 2, 0.3, 0.4, 1500, 3.5) == ([], [])


def test_max_candidates_keeps_the_newest_borders():
    from oracle.cvlike import db_postprocess

    pred = synthetic_prob_map(3)
    q, _ = run_cpp(pred, pred.shape, 2, 0.3, 0.4, 3, 3.5)
    rq, _ = db_postprocess(pred, pred.shape, 2, 0.3, 0.4, 3, 3.5)
    assert q == rq and len(q) <= 3


@pytest.mark.parametrize("seed", range(24))
def test_cpp_matches_oracle_on_noise_fields(seed):
    """Organic shapes: thresholded smooth noise gives blobs with concavities, holes, one-pixel bridges and diagonal
    (8-connected) joints - the cases that separate a correct border follower from an almost correct one."""
    from oracle.cvlike import db_postprocess

    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(40, 110)), int(rng.integers(48, 140))
    field = _blur(_blur(rng.random((h, w)).astype(np.float32), 7), 5)
    field = (field - field.min()) / (field.max() - field.min() + 1e-9)
    pred = np.clip(field * 1.15 - 0.1 + (rng.random((h, w)) > 0.997) * 0.7, 0, 1).astype(np.float32)
    params = (1, float(rng.uniform(0.35, 0.6)), float(rng.uniform(0.3, 0.6)), 1500, float(rng.uniform(1.2, 4.0)))
    size = (int(h * rng.uniform(1.0, 3.0)), int(w * rng.uniform(1.0, 3.0)))
    ref_q, ref_s = db_postprocess(pred, size, *params)
    q, s = run_cpp(pred, size, *params)
    assert q == ref_q
    assert np.allclose(s, ref_s, rtol=0, atol=1e-9)


def test_candidate_cap_and_degenerate_shapes():
    from oracle.cvlike import db_postprocess

    pred = np.zeros((120, 200), dtype=np.float32)
    for i, (y, x) in enumerate((yy, xx) for yy in range(4, 110, 12) for xx in range(4, 190, 16)):
        pred[y : y + 6, x : x + 10] = 0.9  # 108 separate blobs
    for cap in (1500, 40, 1):
        q, s = run_cpp(pred, (240, 400), 2, 0.3, 0.4, cap, 2.0)
        rq, rs = db_postprocess(pred, (240, 400), 2, 0.3, 0.4, cap, 2.0)
        assert q == rq and np.allclose(s, rs)
        assert len(q) <= cap
    thin = np.zeros((30, 60), dtype=np.float32)
    thin[10, 5:50] = 0.9   # one pixel high
    thin[15:28, 55] = 0.9  # one pixel wide
    thin[3, 3] = 0.9       # a single pixel
    q, s = run_cpp(thin, (30, 60), 1, 0.3, 0.1, 1500, 2.0)
    rq, rs = db_postprocess(thin, (30, 60), 1, 0.3, 0.1, 1500, 2.0)
    assert q == rq and np.allclose(s, rs)

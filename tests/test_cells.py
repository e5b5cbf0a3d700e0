"""Table cell detector host logic (yomitoku_amd/table_cell_detector.py): post-processing pinned against the reference's
own functions and CellDetector methods (tests/golden/cells.json, written by oracle/pin_against_reference.py cells), and
the C++ hole finder (ymk_table_hole_rects, the restatement of find_holes_as_rects' OpenCV calls) against an independent
scipy.ndimage formulation.  CPU only: the library is loaded for its host routine, no kernel runs."""
import ctypes
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _detector():
    from yomitoku_amd.layout_parser import RTDETRPostProcessor
    from yomitoku_amd.table_cell_detector import CellDetector

    det = object.__new__(CellDetector)  # the constructor wants a HIP device; the post-processing does not
    det.postprocessor = RTDETRPostProcessor(num_classes=6, num_top_queries=1500)
    det.thresh_score = 0.5
    det.label_mapper = dict(enumerate(["table", "cell", "header", "empty", "kv_item", "grid"]))
    return det


def test_postprocess_matches_reference_golden():
    with open(os.path.join(GOLD, "cells.json")) as f:
        cases = json.load(f)["cases"]
    det = _detector()
    adopted = 0
    for case in cases:
        n = case["n_real"]
        lg = np.concatenate([np.asarray(case["logits"], dtype=np.float32).reshape(n, 6), np.full((1500 - n, 6), -8.0, dtype=np.float32)])
        bx = np.concatenate([np.asarray(case["boxes"], dtype=np.float32).reshape(n, 4), np.full((1500 - n, 4), 0.2, dtype=np.float32)])
        h, w = case["size"]
        cells, kv, grid = det.postprocess({"pred_logits": lg[None], "pred_boxes": bx[None]},
                                          {"size": (h, w), "offset": tuple(case["offset"])}, list(case["table_box"]))
        assert [c.model_dump() for c in cells] == case["cells"]
        got_kv, got_grid = [r.model_dump() for r in kv], [r.model_dump() for r in grid]
        for got, want in ((got_kv, case["kv"]), (got_grid, case["grid"])):
            assert [(g["box"], g["role"]) for g in got] == [(g["box"], g["role"]) for g in want]
            assert np.allclose([g["score"] for g in got], [g["score"] for g in want], rtol=1e-6)
        adopted += len(cells)
    assert adopted > 300


def _holes(h, w, boxes, pad=2, ks=5, min_area=300):
    from yomitoku_amd.table_cell_detector import find_holes_as_rects

    return find_holes_as_rects((h, w), boxes, pad, ks, min_area)


def _holes_scipy(h, w, boxes, pad=2, ks=5, min_area=300):
    """The same pipeline with scipy.ndimage: filled rectangles (inclusive, clipped), OPEN by a box x 3 iterations,
    corner flood fill (4-connected), 8-connected components, bounding boxes."""
    from scipy import ndimage

    m = np.full((h, w), 255, np.uint8)
    for x1, y1, x2, y2 in boxes:
        x1, y1, x2, y2 = max(x1, 0), max(y1, 0), min(x2, w - 1), min(y2, h - 1)
        if x2 >= x1 and y2 >= y1:
            m[y1 : y2 + 1, x1 : x2 + 1] = 0
    k = np.ones((ks, ks), bool)
    white = m != 0
    white = ndimage.binary_erosion(white, structure=k, iterations=3, border_value=1)
    white = ndimage.binary_dilation(white, structure=k, iterations=3, border_value=0)
    if white[0, 0]:
        lab4, _ = ndimage.label(white, structure=[[0, 1, 0], [1, 1, 1], [0, 1, 0]])
        white &= lab4 != lab4[0, 0]
    lab, _ = ndimage.label(white, structure=np.ones((3, 3)))
    rects = []
    for sl in ndimage.find_objects(lab):
        y0, y1, x0, x1 = sl[0].start, sl[0].stop, sl[1].start, sl[1].stop
        if (x1 - x0) * (y1 - y0) >= min_area:
            rects.append([x0 - pad, y0 - pad, x1 + pad, y1 + pad])
    return rects


def test_hole_rects_grid_with_a_missing_cell():
    cells = [[c * 100, r * 60, c * 100 + 99, r * 60 + 59] for r in range(3) for c in range(3) if (r, c) != (1, 1)]
    assert _holes(180, 300, cells) == [[98, 58, 202, 122]]
    assert _holes(180, 300, []) == []  # all white: connected to the corner, flooded away
    assert _holes(50, 60, [[0, 0, 59, 49]]) == []


@pytest.mark.parametrize("seed", range(4))
def test_hole_rects_vs_scipy(seed):
    """Proves the mask arithmetic (inclusive fills, 13 x 13 opening with a border that never wins, corner flood fill,
    8-connected components, padded bounding boxes).  Cannot prove OpenCV's contour ORDER (compared as sets) nor its
    RETR_EXTERNAL nesting rule on components enclosed by other components (none arise from rectangle unions here)."""
    rng = np.random.default_rng(seed)
    for _ in range(60):
        h, w = int(rng.integers(60, 400)), int(rng.integers(60, 500))
        boxes = []
        for _b in range(int(rng.integers(0, 25))):
            x, y = int(rng.integers(-10, w)), int(rng.integers(-10, h))
            boxes.append([x, y, x + int(rng.integers(5, 150)), y + int(rng.integers(5, 100))])
        assert sorted(map(tuple, _holes(h, w, boxes))) == sorted(map(tuple, _holes_scipy(h, w, boxes)))


def test_hole_rects_capacity_grows():
    # 40 x 40 isolated 12 x 12 holes: more than the first capacity guess of the wrapper
    boxes = []
    step, n = 30, 20
    for r in range(n + 1):
        boxes.append([0, r * step, n * step + 17, r * step + 17])
        boxes.append([r * step, 0, r * step + 17, n * step + 17])
    rects = _holes(n * step + 18, n * step + 18, boxes, pad=0, ks=1, min_area=1)
    assert len(rects) == n * n


def test_adjacency_predicates_basic():
    from yomitoku_amd.geometry import calc_iou, is_bottom_adjacent, is_right_adjacent

    a, b = [0, 0, 100, 50], [103, 2, 200, 52]
    assert is_right_adjacent(a, b) and not is_right_adjacent(b, a)
    assert is_bottom_adjacent([0, 0, 100, 50], [5, 53, 110, 100]) and not is_bottom_adjacent([0, 0, 100, 50], [150, 53, 250, 100])
    assert calc_iou([0, 0, 10, 10], [0, 0, 10, 10]) == 1.0 and calc_iou([0, 0, 10, 10], [20, 20, 30, 30]) == 0
    for rule in ("hard", "soft", "nest", "child"):
        assert isinstance(is_right_adjacent(a, b, rule=rule), bool) and isinstance(is_bottom_adjacent(a, b, rule=rule), bool)


def test_adjacency_matrices_equal_the_scalar_predicates():
    """The matrix form behind calc_adjacent_holes_to_cells against the scalar restatement of utils/misc.py:299-441 (itself
    pinned through tests/golden/cells.json) on random table-like boxes, touching, nested and degenerate ones included."""
    from yomitoku_amd.geometry import adjacency_matrices, is_bottom_adjacent, is_right_adjacent

    rng = np.random.default_rng(11)
    for trial in range(30):
        n, m = int(rng.integers(1, 14)), int(rng.integers(1, 14))
        def boxes(k):
            x1, y1 = rng.integers(0, 300, k), rng.integers(0, 300, k)
            w, h = rng.integers(0 if trial % 5 == 0 else 4, 120, k), rng.integers(0 if trial % 5 == 0 else 4, 80, k)
            return np.stack([x1, y1, x1 + w, y1 + h], 1).tolist()
        a, b = boxes(n), boxes(m)
        if trial % 3 == 0:  # make some exact neighbours: b_j starts where a_i ends
            for i in range(min(n, m)):
                b[i] = [a[i][2] + int(rng.integers(0, 20)), a[i][1] + int(rng.integers(-5, 6)), a[i][2] + 60, a[i][3] + int(rng.integers(-5, 6))]
        right, bottom = adjacency_matrices(a, b)
        for i in range(n):
            for j in range(m):
                assert bool(right[i][j]) == is_right_adjacent(a[i], b[j]), (a[i], b[j])
                assert bool(bottom[i][j]) == is_bottom_adjacent(a[i], b[j]), (a[i], b[j])
    r, d = adjacency_matrices([], [[0, 0, 1, 1]])
    assert r.shape == (0, 1) and d.shape == (0, 1)


def test_choose_role_ties():
    from yomitoku_amd.table_cell_detector import choose_role

    assert choose_role({}) is None
    assert choose_role({"cell": 2, "header": 2, "empty": 0}) == "cell"
    assert choose_role({"cell": 0, "header": 3, "empty": 3}) == "header"
    assert choose_role({"cell": 1, "header": 0, "empty": 4}) == "empty"
    assert choose_role({"cell": 0, "header": 0, "empty": 0}) == "cell"

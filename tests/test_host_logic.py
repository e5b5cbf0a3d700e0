"""CPU parity of the pure-Python stages against answers the REFERENCE functions produced
(oracle/pin_against_reference.py host -> tests/golden/host_logic.json): reading order must be
bit-exact (north_star), box predicates exact."""
import json
import os
from types import SimpleNamespace

import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "host_logic.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


@pytest.mark.parametrize("vector_min", [2, 6, 10 ** 9], ids=["numpy-graph", "default", "scalar-graph"])
def test_reading_order_matches_reference(gold, vector_min, monkeypatch):
    from yomitoku_amd import reading_order as ro

    monkeypatch.setattr(ro, "VECTOR_MIN_BOXES", vector_min)  # both graph builders must give the reference's order
    assert len(gold["reading_order"]) >= 200
    for case in gold["reading_order"]:
        els = [SimpleNamespace(box=b, order=0) for b in case["boxes"]]
        ro.prediction_reading_order(els, case["direction"])
        assert [e.order for e in els] == case["order"], case["direction"]


def test_vector_graph_equals_scalar_graph_on_dense_pages():
    """Pages far bigger than the golden cases: the numpy edge list must equal the scalar double loop's."""
    import random

    from yomitoku_amd import reading_order as ro

    rng = random.Random(5)
    for trial in range(30):
        n = rng.randint(6, 90)
        boxes = []
        for _ in range(n):
            x, y = rng.randint(0, 1500), rng.randint(0, 1100)
            boxes.append([x, y, x + rng.randint(1, 400), y + rng.randint(1, 120)])
        for direction in ("top2bottom", "right2left", "left2right"):
            fast = ro._build([list(b) for b in boxes], direction)
            ro.VECTOR_MIN_BOXES, keep = 10 ** 9, ro.VECTOR_MIN_BOXES
            try:
                slow = ro._build([list(b) for b in boxes], direction)
            finally:
                ro.VECTOR_MIN_BOXES = keep
            assert fast.children == slow.children and fast.distance == slow.distance
            assert [sorted(p) for p in fast.parents] == [sorted(p) for p in slow.parents]


def test_containment_matrix_equals_scalar_predicate(gold):
    import numpy as np

    from yomitoku_amd import geometry as g

    a = [p["a"] for p in gold["pairs"]]
    b = [p["b"] for p in gold["pairs"]]
    for thr in (0.5, 0.7, 0.8):
        m = g.containment_matrix(a, b, thr)
        want = np.array([[g.is_contained(x, y, thr) for y in b] for x in a])
        assert (m == want).all()
    assert g.containment_matrix([], b, 0.5).shape == (0, len(b))


def test_reading_order_small_inputs():
    from yomitoku_amd.reading_order import prediction_reading_order

    assert prediction_reading_order([], "top2bottom") == []
    one = [SimpleNamespace(box=[0, 0, 10, 10], order=7)]
    assert prediction_reading_order(one, "left2right")[0].order == 7  # untouched below 2 elements
    with pytest.raises(ValueError):
        prediction_reading_order([SimpleNamespace(box=[0, 0, 1, 1], order=0)] * 2, "diagonal")


def test_box_predicates_match_reference(gold):
    from yomitoku_amd import geometry as g

    for p in gold["pairs"]:
        ratio, inter = g.calc_overlap_ratio(p["a"], p["b"])
        assert ratio == pytest.approx(p["ratio"], abs=0) and inter == p["inter"]
        assert g.is_contained(p["a"], p["b"]) == p["contained"]
        assert g.is_intersected_horizontal(p["a"], p["b"]) == p["ih"]
        assert g.is_intersected_vertical(p["a"], p["b"]) == p["iv"]


def test_quad_to_xyxy_and_filter():
    from yomitoku_amd.geometry import filter_by_flag, quad_to_xyxy

    assert quad_to_xyxy([[5, 9], [40, 7], [41, 30], [4, 31]]) == (4, 7, 41, 31)
    assert filter_by_flag(["a", "b", "c"], [True, False, True]) == ["a", "c"]


def test_reading_order_is_a_permutation_at_page_scale():
    """Size-independent property at sizes far beyond the golden cases: every element gets a distinct rank 0..n-1,
    whatever the layout, and ranking is a pure function of the boxes (second call, same answer)."""
    import random

    from yomitoku_amd.reading_order import prediction_reading_order

    rng = random.Random(11)
    for n in (2, 7, 64, 300):
        for direction in ("top2bottom", "right2left", "left2right"):
            boxes = []
            for _ in range(n):
                x, y = rng.randint(0, 1500), rng.randint(0, 2000)
                boxes.append([x, y, x + rng.randint(2, 500), y + rng.randint(2, 90)])
            els = [SimpleNamespace(box=list(b), order=0) for b in boxes]
            prediction_reading_order(els, direction)
            first = [e.order for e in els]
            assert sorted(first) == list(range(n))
            again = [SimpleNamespace(box=list(b), order=0) for b in boxes]
            prediction_reading_order(again, direction)
            assert [e.order for e in again] == first

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


def test_reading_order_matches_reference(gold):
    from yomitoku_amd.reading_order import prediction_reading_order

    assert len(gold["reading_order"]) >= 200
    for case in gold["reading_order"]:
        els = [SimpleNamespace(box=b, order=0) for b in case["boxes"]]
        prediction_reading_order(els, case["direction"])
        assert [e.order for e in els] == case["order"], case["direction"]


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

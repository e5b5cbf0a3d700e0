"""oracle/hostlogic.py - the oracle's own statement of the reference's host logic (box filters, cell grid, reading order,
aggregation) - pinned against the answers the REFERENCE's functions gave: the golden files written by
oracle/pin_against_reference.py from the imported reference classes (the same files the product's mirrors are held to,
tests/test_{aggregate,filters,host_logic}.py).  Integer / string data: equality is exact.  With these green,
oracle.pipeline.analyze is a free-running CPU DocumentAnalyzer that shares no line with yomitoku_amd."""
import copy
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _gold(name):
    with open(os.path.join(GOLD, name), encoding="utf-8") as f:
        return json.load(f)


def test_nothing_of_the_product_is_imported():
    import ast

    import oracle.hostlogic as hl

    tree = ast.parse(open(hl.__file__, encoding="utf-8").read())
    mods = {n.module for n in ast.walk(tree) if isinstance(n, ast.ImportFrom)} | {a.name for n in ast.walk(tree) if isinstance(n, ast.Import) for a in n.names}
    assert mods <= {"__future__", "math", "re"}, mods


def test_rectangle_predicates_match_reference():
    from oracle import hostlogic as hl

    pairs = _gold("host_logic.json")["pairs"]
    assert len(pairs) >= 200
    for p in pairs:
        ratio, inter = hl.overlap_of_b(p["a"], p["b"])
        assert ratio == p["ratio"] and inter == p["inter"]
        assert hl.contains(p["a"], p["b"]) == p["contained"]
        assert hl.rows_overlap(p["a"], p["b"]) == p["ih"]
        assert hl.columns_overlap(p["a"], p["b"]) == p["iv"]
    assert hl.quad_box([[5, 9], [40, 7], [41, 30], [4, 31]]) == [4, 7, 41, 31]


def test_reading_order_matches_reference():
    from oracle import hostlogic as hl

    cases = _gold("host_logic.json")["reading_order"]
    assert len(cases) >= 200
    for case in cases:
        els = [{"box": list(b), "order": 0} for b in case["boxes"]]
        hl.reading_order(els, case["direction"])
        assert [e["order"] for e in els] == case["order"], case["direction"]
    one = [{"box": [0, 0, 10, 10], "order": 7}]
    assert hl.reading_order(one, "left2right")[0]["order"] == 7 and hl.reading_order([], "top2bottom") == []
    with pytest.raises(ValueError):
        hl.reading_order([{"box": [0, 0, 1, 1], "order": 0}] * 2, "diagonal")


def test_layout_filters_and_cell_grid_match_reference():
    from oracle import hostlogic as hl

    gold = _gold("filters.json")
    for case in gold["layout"]:
        a = hl.drop_nested_within_category(copy.deepcopy(case["input"]))
        assert a == case["within"]
        b = hl.drop_targets_inside_sources(copy.deepcopy(a), "tables", "paragraphs")
        assert b == case["after_tables"]
        assert hl.drop_targets_inside_sources(copy.deepcopy(b), "figures", "paragraphs") == case["after_figures"]
    merged_any = False
    for case in gold["table"]:
        cells = hl.grid_cells(case["rows"], case["cols"])
        assert cells == case["cells"]
        merged = hl.merge_span_cells(copy.deepcopy(cells), case["spans"])
        assert merged == case["merged"]
        merged_any |= any(c["row_span"] > 1 or c["col_span"] > 1 for c in merged)
    assert merged_any


def test_rtdetr_post_matches_reference():
    """oracle.pipeline.rtdetr_post (what analyze() feeds the filters with) against the reference RTDETRPostProcessor's answers."""
    from oracle.pipeline import rtdetr_post

    for case in _gold("filters.json")["post"]:
        out = rtdetr_post(np.asarray(case["logits"], dtype=np.float32)[None], np.asarray(case["boxes"], dtype=np.float32)[None],
                          tuple(case["size_wh"]), case["threshold"], case["num_classes"])[0]
        assert out["labels"].tolist() == case["labels"]
        np.testing.assert_allclose(out["scores"], np.asarray(case["scores"], dtype=np.float32), rtol=1e-6, atol=0)
        np.testing.assert_allclose(out["boxes"], np.asarray(case["out_boxes"], dtype=np.float32), rtol=1e-6, atol=1e-4)


def test_aggregate_matches_reference():
    from oracle import hostlogic as hl

    cases = _gold("aggregate.json")
    assert len(cases) >= 50
    seen = {"tables": 0, "figures": 0, "ruby": 0, "vertical": 0}
    for case in cases:
        opts = case["input"]["opts"]
        words = copy.deepcopy(case["input"]["ocr"]["words"])
        layout = copy.deepcopy(case["input"]["layout"])
        out = hl.aggregate(words, layout, ignore_meta=opts["ignore_meta"], reading_order_opt=opts["reading_order"],
                           ignore_ruby=opts["ignore_ruby"], ruby_threshold=opts["ruby_threshold"])
        assert out == case["output"]
        seen["tables"] += len(out["tables"])
        seen["figures"] += sum(len(f["paragraphs"]) for f in out["figures"])
        seen["ruby"] += int(opts["ignore_ruby"])
        seen["vertical"] += sum(p["direction"] == "vertical" for p in out["paragraphs"])
    assert all(v > 0 for v in seen.values()), seen  # the cases do reach every branch


def test_split_text_across_cells_matches_reference():
    from oracle import hostlogic as hl

    cut = untouched = 0
    for case in _gold("aggregate.json"):
        words = case["input"]["ocr"]["words"]
        points, scores = [w["points"] for w in words], [w["det_score"] for w in words]
        got_p, got_s = hl.split_text_across_cells(copy.deepcopy(points), list(scores), case["input"]["layout"]["tables"])
        assert {"points": got_p, "scores": got_s} == case["split"]
        cut += sum(1 for q in got_p if q not in points)
        untouched += sum(1 for q in got_p if q in points)
    assert cut > 20 and untouched > 100, (cut, untouched)  # the cases do cut words at cell borders

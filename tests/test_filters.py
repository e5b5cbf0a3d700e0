"""Layout / table box filters and the RT-DETR output post-processor against answers of the REFERENCE's own
functions (oracle/pin_against_reference.py filters -> tests/golden/filters.json).  Integer box logic: exact;
scores / scaled boxes: fp32, compared to 1e-6 relative."""
import copy
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "filters.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_layout_filters_match_reference(gold):
    from yomitoku_amd.layout_parser import (filter_contained_rectangles_across_categories,
                                            filter_contained_rectangles_within_category)

    assert len(gold["layout"]) >= 50
    dropped = 0
    for case in gold["layout"]:
        a = filter_contained_rectangles_within_category(copy.deepcopy(case["input"]))
        assert a == case["within"]
        b = filter_contained_rectangles_across_categories(copy.deepcopy(a), "tables", "paragraphs")
        assert b == case["after_tables"]
        c = filter_contained_rectangles_across_categories(copy.deepcopy(b), "figures", "paragraphs")
        assert c == case["after_figures"]
        dropped += sum(len(v) for v in case["input"].values()) - sum(len(v) for v in c.values())
    assert dropped > 50  # the cases do exercise the filters


def test_table_cell_grid_matches_reference(gold):
    from yomitoku_amd.table_structure_recognizer import extract_cells, filter_contained_cells_within_spancell

    merged_any = False
    for case in gold["table"]:
        cells = extract_cells(case["rows"], case["cols"])
        assert cells == case["cells"]
        merged = filter_contained_cells_within_spancell(copy.deepcopy(cells), case["spans"])
        assert merged == case["merged"]
        merged_any |= any(c["row_span"] > 1 or c["col_span"] > 1 for c in merged)
    assert merged_any


def test_rtdetr_postprocessor_matches_reference(gold):
    from yomitoku_amd.layout_parser import RTDETRPostProcessor

    for case in gold["post"]:
        post = RTDETRPostProcessor(num_classes=case["num_classes"], num_top_queries=300)
        out = post({"pred_logits": np.asarray(case["logits"], dtype=np.float32)[None],
                    "pred_boxes": np.asarray(case["boxes"], dtype=np.float32)[None]}, case["size_wh"], case["threshold"])[0]
        assert 0 < len(case["labels"]) < 300
        assert out["labels"].tolist() == case["labels"]
        np.testing.assert_allclose(out["scores"], np.asarray(case["scores"], dtype=np.float32), rtol=1e-6, atol=0)
        np.testing.assert_allclose(out["boxes"], np.asarray(case["out_boxes"], dtype=np.float32), rtol=1e-6, atol=1e-4)

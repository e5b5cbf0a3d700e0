"""CPU parity of the page aggregation (word -> cell / paragraph assignment, furigana filter, figure
capture, header / footer split, reading order) and of split_text_across_cells against answers
produced by the REFERENCE's DocumentAnalyzer.aggregate / _split_text_across_cells
(oracle/pin_against_reference.py aggregate -> tests/golden/aggregate.json).  Everything here is
integer / string data: equality is exact."""
import json
import os
from types import SimpleNamespace

import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "aggregate.json")


@pytest.fixture(scope="module")
def cases():
    with open(GOLD, encoding="utf-8") as f:
        return json.load(f)


def test_aggregate_matches_reference(cases):
    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd import schemas as sch

    assert len(cases) >= 50
    for case in cases:
        ocr = sch.OCRSchema(**case["input"]["ocr"])
        layout = sch.LayoutAnalyzerSchema(**case["input"]["layout"])
        me = SimpleNamespace(img=None, **case["input"]["opts"])
        out = sch.DocumentAnalyzerSchema(**da.DocumentAnalyzer.aggregate(me, ocr, layout)).model_dump()
        assert out == case["output"]


def test_split_text_across_cells_matches_reference(cases):
    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd import schemas as sch

    for case in cases:
        ocr = case["input"]["ocr"]
        det = sch.TextDetectorSchema(points=[w["points"] for w in ocr["words"]], scores=[w["det_score"] for w in ocr["words"]])
        layout = sch.LayoutAnalyzerSchema(**case["input"]["layout"])
        assert da._split_text_across_cells(det, layout).model_dump() == case["split"]


def test_known_answers_from_the_reference_unit_tests():
    """Spot values the reference's own tests pin (tests/test_document_analyzer.py:116-607)."""
    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd import schemas as sch

    assert da.combine_flags([True, False, False], [False, False, True]) == [True, False, True]
    assert da.is_vertical([[0, 0], [10, 0], [10, 30], [0, 30]]) and not da.is_vertical([[0, 0], [30, 0], [30, 10], [0, 10]])
    assert da.is_noise([[0, 0], [10, 0], [10, 30], [0, 30]]) and not da.is_noise([[0, 0], [20, 0], [20, 30], [0, 30]])
    assert da.recursive_update({"a": {"b": 1, "c": 2}, "d": 3}, {"a": {"b": 9}, "e": 4}) == {"a": {"b": 9, "c": 2}, "d": 3, "e": 4}
    hp = [sch.ParagraphSchema(box=[0, 0, 100, 10], contents="x", direction="horizontal", order=0, role=None)]
    vp = [sch.ParagraphSchema(box=[0, 0, 10, 200], contents="y", direction="vertical", order=0, role=None)]
    assert da.judge_page_direction(hp) == "horizontal" and da.judge_page_direction(hp + vp) == "vertical"
    line = lambda b: SimpleNamespace(box=b)  # noqa: E731
    words = [{"points": [[0, 0], [10, 0], [10, 10], [0, 10]]}, {"points": [[0, 5], [10, 5], [10, 15], [0, 15]]}]
    assert da._calc_overlap_words_on_lines([line([0, 0, 10, 10]), line([0, 20, 10, 30])], words) == [[1.0, 0.0], [0.5, 0.0]]


def test_configs_must_be_dicts():
    from yomitoku_amd.document_analyzer import OCR, DocumentAnalyzer, LayoutAnalyzer

    for cls in (OCR, LayoutAnalyzer, DocumentAnalyzer):
        with pytest.raises(ValueError):
            cls(configs="not-a-dict")


def test_missing_config_file_is_reported_before_anything_is_built():
    """tests/test_document_analyzer.py::test_invalid_path, tests/test_ocr.py::test_ocr_invalid_path of the reference."""
    from yomitoku_amd.document_analyzer import OCR, DocumentAnalyzer, LayoutAnalyzer

    with pytest.raises(FileNotFoundError):
        DocumentAnalyzer(configs={"ocr": {"text_detector": {"path_cfg": "tests/yaml/dummy.yaml"}}})
    with pytest.raises(FileNotFoundError):
        OCR(configs={"text_detector": {"path_cfg": "tests/yaml/dummy.yaml"}})
    with pytest.raises(FileNotFoundError):
        LayoutAnalyzer(configs={"layout_parser": {"path_cfg": "tests/yaml/dummy.yaml"}})


def test_import_paths_of_the_reference_resolve():
    """from yomitoku import X / from yomitoku.ocr import OCR / from yomitoku.layout_analyzer import LayoutAnalyzer."""
    import yomitoku_amd
    from yomitoku_amd import document_analyzer
    from yomitoku_amd.layout_analyzer import LayoutAnalyzer
    from yomitoku_amd.ocr import OCR

    assert OCR is document_analyzer.OCR and LayoutAnalyzer is document_analyzer.LayoutAnalyzer
    for name in ("OCR", "LayoutParser", "TableStructureRecognizer", "TextDetector", "TextRecognizer", "LayoutAnalyzer",
                 "DocumentAnalyzer"):
        assert getattr(yomitoku_amd, name).__name__ == name
    assert isinstance(yomitoku_amd.__version__, str)


def test_aggregate_accounts_for_every_word_on_a_full_size_page():
    """Property at BASELINE page scale (1600x1200, ~80 lines, ruled tables): after aggregation every recognised word
    is in exactly one place - a table cell, a paragraph, or on its own - and paragraph ranks are a permutation."""
    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd import schemas as sch
    from yomitoku_amd.utils.synth import synthetic_page_with_truth

    for seed in (0, 5, 9):
        img, quads, tables, paragraphs = synthetic_page_with_truth(seed)
        words = [sch.WordPrediction(points=[[int(a), int(b)] for a, b in q], content=f"w{i}", direction="horizontal",
                                    rec_score=0.9, det_score=0.9) for i, q in enumerate(quads)]
        tabs = []
        for x1, y1, x2, y2 in tables:
            cells = [sch.TableCellSchema(col=c + 1, row=r + 1, col_span=1, row_span=1, contents=None,
                                         box=[x1 + (x2 - x1) * c // 3, y1 + (y2 - y1) * r // 4, x1 + (x2 - x1) * (c + 1) // 3,
                                              y1 + (y2 - y1) * (r + 1) // 4]) for r in range(4) for c in range(3)]
            tabs.append(sch.TableStructureRecognizerSchema(box=[x1, y1, x2, y2], n_row=4, n_col=3, rows=[], cols=[], spans=[],
                                                           cells=cells, order=0))
        lay = sch.LayoutAnalyzerSchema(paragraphs=[sch.Element(id=None, box=list(b), score=1.0, role=None, contents=None)
                                                   for b in paragraphs], tables=tabs, figures=[])
        an = da.DocumentAnalyzer.__new__(da.DocumentAnalyzer)
        an.reading_order, an.ignore_meta, an.ignore_ruby, an.ruby_threshold, an.img = "auto", False, False, 2.0, img
        out = an.aggregate(sch.OCRSchema(words=words), lay)
        placed = []
        for t in out["tables"]:
            for c in t.cells:
                placed += [w for w in c.contents.split("\n") if w]
        for p in out["paragraphs"]:
            placed += p.contents.split("\n")
        assert sorted(placed) == sorted(w.content for w in words)  # each word exactly once
        ranks = sorted([p.order for p in out["paragraphs"]] + [t.order for t in out["tables"]])
        assert ranks == list(range(len(ranks)))

"""Exporters (yomitoku_amd/export.py) against the reference's own conversion functions (tests/golden/export.json, written by
oracle/pin_against_reference.py export), plus the file-level behaviour of the four export_* entry points."""
import csv
import json
import os

import numpy as np

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _doc(d):
    from yomitoku_amd.schemas import DocumentAnalyzerSchema

    return DocumentAnalyzerSchema(**d)


def test_text_conversions_match_reference_golden():
    from yomitoku_amd import export as ex

    with open(os.path.join(GOLD, "export.json"), encoding="utf-8") as f:
        cases = json.load(f)["cases"]
    assert len(cases) >= 40
    for c in cases:
        doc, ilb, letter = _doc(c["doc"]), c["ignore_line_break"], c["figure_letter"]
        got = ex.convert_csv(doc.model_copy(deep=True), "out.csv", ilb, None, False, letter, "figures")
        assert [{"type": e["type"], "element": e["element"], "order": e["order"]} for e in got] == c["csv"]
        md, _ = ex.convert_markdown(doc.model_copy(deep=True), "out.md", ilb, None, letter, False, 200, "figures")
        assert md == c["markdown"]
        assert [ex.table_to_html(t, ilb)["html"] for t in doc.tables] == c["html_tables"]
        assert [ex.paragraph_to_html(p, ilb)["html"] for p in doc.paragraphs] == c["html_paragraphs"]


def test_export_entry_points_write_files(tmp_path):
    from yomitoku_amd import export as ex

    with open(os.path.join(GOLD, "export.json"), encoding="utf-8") as f:
        doc = _doc(next(c["doc"] for c in json.load(f)["cases"] if c["doc"]["figures"] and c["doc"]["tables"]))
    img = np.random.default_rng(0).integers(0, 256, size=(64, 64, 3), dtype=np.uint8)
    out = doc.model_copy(deep=True).to_json(str(tmp_path / "a" / "page.json"), img=img, export_figure=True) if False else None
    os.makedirs(tmp_path / "a", exist_ok=True)
    res = ex.export_json(doc.model_copy(deep=True), str(tmp_path / "a" / "page.json"), img=img, export_figure=True)
    data = json.load(open(tmp_path / "a" / "page.json", encoding="utf-8"))
    assert data["figures"][0]["figure_path"] == os.path.join("figures", "page_figure_0.png")
    assert os.path.getsize(tmp_path / "a" / "figures" / "page_figure_0.png") > 0 and res.figures[0].figure_path
    assert list(data) == sorted(data)  # sort_keys=True, indent 4
    ex.export_csv(doc.model_copy(deep=True), str(tmp_path / "a" / "page.csv"), img=img)
    rows = list(csv.reader(open(tmp_path / "a" / "page.csv", newline="", encoding="utf-8")))
    assert [""] in rows and len(rows) > len(doc.paragraphs)
    md = doc.model_copy(deep=True).to_markdown(str(tmp_path / "a" / "page.md"), img=img)
    assert '<img src="figures/page_figure_0.png" width="200px"><br>' in md and open(tmp_path / "a" / "page.md", encoding="utf-8").read() == md
    html = doc.model_copy(deep=True).to_html(str(tmp_path / "a" / "page.html"), img=img, export_figure_letter=True)
    assert "<table border=" in html and 'figures/page_figure_0.png" width="200"' in html
    empty = _doc({"paragraphs": [], "tables": [], "words": [], "figures": []})
    assert ex.export_html(empty, str(tmp_path / "e.html"), export_figure=False) == ""
    assert ex.export_markdown(empty, str(tmp_path / "e.md"), export_figure=False) == ""

"""Exporters of a DocumentAnalyzerSchema - the step after the path (reference export/*.py): JSON,
CSV, Markdown and HTML, with the reference's function names, arguments and output text.  Pure host code.

Two dependencies of the reference are not requirements here:
  - figure crops are written with Pillow (the reference calls cv2.imencode(".jpg") - yes, JPEG bytes under a .png name,
    utils/misc.py:21-32; kept: JPEG data, the same file names);
  - the HTML string is pretty-printed with lxml when lxml is installed (export_html.py:174-179) and returned as built
    otherwise (same elements, no re-indentation).
The text conversion functions are pinned against the reference's own (tests/golden/export.json)."""

from __future__ import annotations

import csv
import json
import os
import re
from html import escape


def save_image(img, path):
    """utils/misc.py:21-32: JPEG-encoded bytes written to `path` (BGR input)."""
    from PIL import Image

    basedir = os.path.dirname(path)
    if basedir:
        os.makedirs(basedir, exist_ok=True)
    import io

    import numpy as np

    buf = io.BytesIO()
    arr = np.ascontiguousarray(np.asarray(img)[:, :, ::-1]) if np.asarray(img).ndim == 3 else np.asarray(img)
    try:
        Image.fromarray(arr).save(buf, format="JPEG", quality=95)
    except Exception as exc:
        raise ValueError("Failed to encode image") from exc
    with open(path, "wb") as f:
        f.write(buf.getvalue())


def _save_figures(figures, img, out_path, figure_dir):
    """The figure crops next to `out_path`; yields (index, figure, name relative to figure_dir)."""
    assert img is not None, "img is required for saving figures"
    out = []
    for i, figure in enumerate(figures):
        x1, y1, x2, y2 = map(int, figure.box)
        save_dir = os.path.join(os.path.dirname(out_path), figure_dir)
        os.makedirs(save_dir, exist_ok=True)
        name = f"{os.path.splitext(os.path.basename(out_path))[0]}_figure_{i}.png"
        save_image(img[y1:y2, x1:x2, :], os.path.join(save_dir, name))
        out.append((i, figure, name))
    return out


# ---------------------------------------------------------------------------------------------- JSON (export_json.py)
def paragraph_to_json(paragraph, ignore_line_break):
    """In place (export_json.py:7): the line breaks of a paragraph's text go when asked."""
    if ignore_line_break:
        paragraph.contents = paragraph.contents.replace("\n", "")


def table_to_json(table, ignore_line_break):
    """In place (export_json.py:12): the same for every cell of a table."""
    if ignore_line_break:
        for cell in table.cells:
            cell.contents = cell.contents.replace("\n", "")


def convert_json(inputs, out_path, ignore_line_break, img, export_figure, figure_dir):
    from .schemas import DocumentAnalyzerSchema

    if isinstance(inputs, DocumentAnalyzerSchema):
        for table in inputs.tables:
            table_to_json(table, ignore_line_break)
        for paragraph in inputs.paragraphs:
            paragraph_to_json(paragraph, ignore_line_break)
        if export_figure:
            for _, figure, name in _save_figures(inputs.figures, img, out_path, figure_dir):
                figure.figure_path = os.path.join(figure_dir, name)
    return inputs


def save_json(data, out_path, encoding):
    with open(out_path, "w", encoding=encoding, errors="ignore") as f:
        json.dump(data, f, ensure_ascii=False, indent=4, sort_keys=True, separators=(",", ": "))


def export_json(inputs, out_path, ignore_line_break=False, encoding: str = "utf-8", img=None, export_figure=False,
                figure_dir="figures"):
    inputs = convert_json(inputs, out_path, ignore_line_break, img, export_figure, figure_dir)
    save_json(inputs.model_dump(), out_path, encoding)
    return inputs


# ---------------------------------------------------------------------------------------------- CSV (export_csv.py)
def table_to_csv(table, ignore_line_break):
    grid = [["" for _ in range(table.n_col)] for _ in range(table.n_row)]
    for cell in table.cells:
        contents = cell.contents
        if ignore_line_break:
            contents = contents.replace("\n", "")
        grid[cell.row - 1][cell.col - 1] = contents  # only the top-left slot of a spanning cell carries the text
    return grid


def paragraph_to_csv(paragraph, ignore_line_break):
    return paragraph.contents.replace("\n", "") if ignore_line_break else paragraph.contents


def convert_csv(inputs, out_path, ignore_line_break, img=None, export_figure: bool = True, export_figure_letter: bool = False,
                figure_dir="figures"):
    elements = [{"type": "table", "box": t.box, "element": table_to_csv(t, ignore_line_break), "order": t.order} for t in inputs.tables]
    elements += [{"type": "paragraph", "box": p.box, "element": paragraph_to_csv(p, ignore_line_break), "order": p.order}
                 for p in inputs.paragraphs]
    if export_figure_letter:
        for figure in inputs.figures:
            for paragraph in sorted(figure.paragraphs, key=lambda x: x.order):
                elements.append({"type": "paragraph", "box": paragraph.box, "element": paragraph_to_csv(paragraph, ignore_line_break),
                                 "order": figure.order})
    elements = sorted(elements, key=lambda x: x["order"])
    if export_figure:
        _save_figures(inputs.figures, img, out_path, figure_dir)
    return elements


def save_csv(elements, out_path, encoding):
    with open(out_path, "w", newline="", encoding=encoding, errors="ignore") as f:
        writer = csv.writer(f, quoting=csv.QUOTE_MINIMAL)
        for element in elements:
            if element["type"] == "table":
                writer.writerows(element["element"])
            else:
                writer.writerow([element["element"]])
            writer.writerow([""])


def export_csv(inputs, out_path: str, ignore_line_break: bool = False, encoding: str = "utf-8", img=None, export_figure: bool = True,
               export_figure_letter: bool = False, figure_dir="figures"):
    elements = convert_csv(inputs, out_path, ignore_line_break, img, export_figure, export_figure_letter, figure_dir)
    save_csv(elements, out_path, encoding)
    return elements


# ---------------------------------------------------------------------------------------------- Markdown (export_markdown.py)
def escape_markdown_special_chars(text):
    return re.sub(r"([`*{}[\]()#+!~|-])", r"\\\1", text)


def _md_text(contents, ignore_line_break):
    contents = escape_markdown_special_chars(contents)
    return contents.replace("\n", "") if ignore_line_break else contents.replace("\n", "<br>")


def paragraph_to_md(paragraph, ignore_line_break):
    contents = _md_text(paragraph.contents, ignore_line_break)
    if paragraph.role == "section_headings":
        contents = "# " + contents
    return {"order": paragraph.order, "box": paragraph.box, "md": contents + "\n"}


def table_to_md(table, ignore_line_break):
    grid = [["" for _ in range(table.n_col)] for _ in range(table.n_row)]
    for cell in table.cells:
        contents = cell.contents
        # the reference escapes once per covered grid slot BEFORE it reaches the top-left one, which it visits first:
        # the text stored is escaped exactly once
        grid[cell.row - 1][cell.col - 1] = _md_text(contents, ignore_line_break)
    md = ""
    for i in range(table.n_row):
        md += "|" + "|".join(grid[i]) + "|\n"
        if i == 0:
            md += "|" + "|".join("-" for _ in range(table.n_col)) + "|\n"
    return {"order": table.order, "box": table.box, "md": md}


def figure_to_md(figures, img, out_path, export_figure_letter=False, ignore_line_break=False, width=200, figure_dir="figures"):
    elements = []
    for _, figure, name in _save_figures(figures, img, out_path, figure_dir):
        elements.append({"order": figure.order, "md": f'<img src="{figure_dir}/{name}" width="{width}px"><br>'})
        if export_figure_letter:
            for paragraph in sorted(figure.paragraphs, key=lambda x: x.order):
                elements.append({"order": figure.order, "md": paragraph_to_md(paragraph, ignore_line_break)["md"]})
    return elements


def convert_markdown(inputs, out_path, ignore_line_break=False, img=None, export_figure_letter=False, export_figure=True,
                     figure_width=200, figure_dir="figures"):
    elements = [table_to_md(t, ignore_line_break) for t in inputs.tables]
    elements += [paragraph_to_md(p, ignore_line_break) for p in inputs.paragraphs]
    if export_figure:
        elements.extend(figure_to_md(inputs.figures, img, out_path, export_figure_letter, ignore_line_break, figure_width, figure_dir=figure_dir))
    elements = sorted(elements, key=lambda x: x["order"])
    return "\n".join(e["md"] for e in elements), elements


def save_markdown(markdown, out_path, encoding):
    with open(out_path, "w", encoding=encoding, errors="ignore") as f:
        f.write(markdown)


def export_markdown(inputs, out_path: str, ignore_line_break: bool = False, img=None, export_figure_letter=False, export_figure=True,
                    figure_width=200, figure_dir="figures", encoding: str = "utf-8"):
    markdown, _ = convert_markdown(inputs, out_path, ignore_line_break, img, export_figure_letter, export_figure, figure_width, figure_dir)
    save_markdown(markdown, out_path, encoding)
    return markdown


# ---------------------------------------------------------------------------------------------- HTML (export_html.py)
def convert_text_to_html(text):
    """HTML-escape the text; URLs stay as plain (escaped) text."""
    return re.compile(r"https?://[^\s<>]").sub(lambda m: escape(m.group(0)), escape(text))


def _html_text(contents, ignore_line_break):
    contents = convert_text_to_html(contents)
    return contents.replace("\n", "") if ignore_line_break else contents.replace("\n", "<br>")


def table_to_html(table, ignore_line_break):
    pre_row, rows, row = 1, [], []
    for cell in table.cells:
        if cell.row != pre_row:
            rows.append("<tr>" + "".join(row) + "</tr>")
            row = []
        contents = _html_text("" if cell.contents is None else cell.contents, ignore_line_break)
        row.append(f'<td rowspan="{cell.row_span}" colspan="{cell.col_span}">{contents}</td>')
        pre_row = cell.row
    rows.append("<tr>" + "".join(row) + "</tr>")
    return {"box": table.box, "order": table.order,
            "html": '<table border="1" style="border-collapse: collapse">' + "".join(rows) + "</table>"}


def paragraph_to_html(paragraph, ignore_line_break):
    contents = _html_text(paragraph.contents, ignore_line_break)
    if paragraph.role == "section_headings":
        contents = f"<h1>{contents}</h1>"
    return {"box": paragraph.box, "order": paragraph.order, "html": f"<p>{contents}</p>"}


def figure_to_html(figures, img, out_path, export_figure_letter=False, ignore_line_break=False, figure_dir="figures", width=200):
    elements = []
    for _, figure, name in _save_figures(figures, img, out_path, figure_dir):
        elements.append({"order": figure.order, "html": f'<img src="{figure_dir}/{name}" width="{width}"><br>'})
        if export_figure_letter:
            for paragraph in sorted(figure.paragraphs, key=lambda x: x.order):
                elements.append({"order": figure.order, "html": paragraph_to_html(paragraph, ignore_line_break)["html"]})
    return elements


def convert_html(inputs, out_path, ignore_line_break, export_figure, export_figure_letter, img=None, figure_width=200,
                 figure_dir="figures"):
    elements = [table_to_html(t, ignore_line_break) for t in inputs.tables]
    elements += [paragraph_to_html(p, ignore_line_break) for p in inputs.paragraphs]
    if export_figure:
        elements.extend(figure_to_html(inputs.figures, img, out_path, export_figure_letter, ignore_line_break, width=figure_width,
                                       figure_dir=figure_dir))
    elements = sorted(elements, key=lambda x: x["order"])
    html_string = "".join(e["html"] for e in elements)
    if html_string:
        try:
            from lxml import etree, html as lxml_html
        except ImportError:
            formatted = html_string  # lxml absent: the elements as built, without re-indentation
        else:
            formatted = etree.tostring(lxml_html.fromstring(html_string), pretty_print=True, encoding="unicode")
    else:
        formatted = ""
    return formatted, elements


def save_html(html, out_path, encoding):
    with open(out_path, "w", encoding=encoding, errors="ignore") as f:
        f.write(html)


def export_html(inputs, out_path: str, ignore_line_break: bool = False, export_figure: bool = True, export_figure_letter: bool = False,
                img=None, figure_width=200, figure_dir="figures", encoding: str = "utf-8"):
    formatted, _ = convert_html(inputs, out_path, ignore_line_break, export_figure, export_figure_letter, img, figure_width, figure_dir)
    save_html(formatted, out_path, encoding)
    return formatted


__all__ = ["export_html", "export_markdown", "export_csv", "export_json", "save_html", "save_markdown", "save_csv", "save_json",
           "convert_html", "convert_markdown", "convert_csv", "convert_json", "save_image"]

"""ORACLE (test infrastructure only - never imported by the product path).

The HOST half of the reference's page analysis restated on plain dicts and lists, element by element as the reference
walks them (no matrices, no pydantic, nothing shared with yomitoku_amd):

  layout_elements   LayoutParser.filtering_elements                     layout_parser.py:30-77, 205-241
  table_structure   TableStructureRecognizer.postprocess                table_structure_recognizer.py:27-85, 210-257
  reading_order     prediction_reading_order + the graph builders        reading_order.py:14-201, utils/graph.py:1-17
  aggregate         DocumentAnalyzer.aggregate and its helpers           document_analyzer.py:19-217, 487-601
  rectangle tests   calc_overlap_ratio ... quad_to_xyxy                  utils/misc.py:35-134

Pinned against the reference itself: tests/test_oracle_hostlogic.py replays the answers the REFERENCE's functions gave
(tests/golden/{aggregate,filters,cells,host_logic}.json, written by oracle/pin_against_reference.py from the imported
reference) on these functions.  With it `oracle.pipeline.analyze` is a free-running CPU statement of
DocumentAnalyzer.__call__ that owes nothing to the product (tools/e2e_oracle_eval.py)."""

from __future__ import annotations

import math
import re

LAYOUT_CATEGORIES = ("tables", "figures", "paragraphs", "section_headings", "page_header", "page_footer")
LAYOUT_ROLES = ("section_headings", "page_header", "page_footer")
TABLE_CATEGORIES = ("row", "col", "span")


# ------------------------------------------------------------------------------------------ rectangles (utils/misc.py)
def intersection(a, b):
    """Integer intersection of two x1 y1 x2 y2 rectangles (coordinates truncated by int()), or None when it is empty."""
    ax1, ay1, ax2, ay2 = (int(v) for v in a)
    bx1, by1, bx2, by2 = (int(v) for v in b)
    x1, y1, x2, y2 = max(ax1, bx1), max(ay1, by1), min(ax2, bx2), min(ay2, by2)
    if max(0, x2 - x1) == 0 or max(0, y2 - y1) == 0:
        return None
    return [x1, y1, x2, y2]


def overlap_of_b(a, b):
    """(share of b's area - from b's own, untruncated coordinates - that the integer intersection covers, intersection)."""
    inter = intersection(a, b)
    if inter is None:
        return 0, None
    area_b = (b[2] - b[0]) * (b[3] - b[1])
    return (inter[2] - inter[0]) * (inter[3] - inter[1]) / area_b, inter


def contains(a, b, threshold=0.8):
    return overlap_of_b(a, b)[0] > threshold


def rows_overlap(a, b, threshold=0.5):
    """is_intersected_horizontal: the vertical extents share at least `threshold` of the shorter one."""
    ay1, ay2, by1, by2 = int(a[1]), int(a[3]), int(b[1]), int(b[3])
    shared = max(0, min(ay2, by2) - max(ay1, by1))
    return not (shared / min(ay2 - ay1, by2 - by1) < threshold)


def columns_overlap(a, b):
    """is_intersected_vertical: the horizontal extents share at least one pixel."""
    return max(0, min(int(a[2]), int(b[2])) - max(int(a[0]), int(b[0]))) != 0


def quad_box(quad):
    xs = [p[0] for p in quad]
    ys = [p[1] for p in quad]
    return [min(xs), min(ys), max(xs), max(ys)]


# ------------------------------------------------------------------------------------------ layout_parser.py:30-77
def drop_nested_within_category(groups):
    """Per category: of two boxes one of which holds (> 80 % of) the other the inner one goes; when each holds the other
    the smaller one goes (the first on equal areas)."""
    out = {}
    for name, items in groups.items():
        keep = [True] * len(items)
        for i in range(len(items)):
            for j in range(i + 1, len(items)):
                bi, bj = items[i]["box"], items[j]["box"]
                i_holds_j, j_holds_i = contains(bi, bj), contains(bj, bi)
                if i_holds_j and j_holds_i:
                    area_i = (bi[2] - bi[0]) * (bi[3] - bi[1])
                    area_j = (bj[2] - bj[0]) * (bj[3] - bj[1])
                    if area_i > area_j:
                        keep[j] = False
                    else:
                        keep[i] = False
                elif i_holds_j:
                    keep[j] = False
                elif j_holds_i:
                    keep[i] = False
        out[name] = [it for it, k in zip(items, keep) if k]
    return out


def drop_targets_inside_sources(groups, source, target):
    keep = [True] * len(groups[target])
    for src in groups[source]:
        for j, tgt in enumerate(groups[target]):
            if contains(src["box"], tgt["box"]):
                keep[j] = False
    groups[target] = [it for it, k in zip(groups[target], keep) if k]
    return groups


def layout_elements(detections, categories=LAYOUT_CATEGORIES, roles=LAYOUT_ROLES):
    """detections: dict(labels, boxes, scores) of oracle.pipeline.rtdetr_post -> {"tables" | "figures" | "paragraphs": [element]}
    (layout_parser.py:205-241: role classes become paragraphs that carry their class as `role`)."""
    groups = {c: [] for c in categories if c not in roles}
    for box, score, label in zip(detections["boxes"], detections["scores"], detections["labels"]):
        category, role = categories[int(label)], None
        if category in roles:
            category, role = "paragraphs", category
        groups[category].append({"id": None, "box": [int(v) for v in box], "score": float(score), "role": role, "contents": None})
    return drop_targets_inside_sources(drop_nested_within_category(groups), "tables", "paragraphs")


# ------------------------------------------------------------------------------------------ table_structure_recognizer.py
def grid_cells(row_boxes, col_boxes):
    cells = []
    for r, rb in enumerate(row_boxes):
        for c, cb in enumerate(col_boxes):
            inter = intersection(rb, cb)
            if inter is not None:
                cells.append({"col": c + 1, "row": r + 1, "col_span": 1, "row_span": 1, "box": inter, "contents": None})
    return cells


def merge_span_cells(cells, span_boxes):
    taken = [False] * len(cells)
    members = []
    for span in span_boxes:
        inside = []
        for j, cell in enumerate(cells):
            if contains(span, cell["box"]):
                taken[j] = True
                inside.append(cell)
        members.append(inside)
    cells = [c for c, t in zip(cells, taken) if not t]
    for span, inside in zip(span_boxes, members):
        if not inside:
            continue
        r0, c0 = min(c["row"] for c in inside), min(c["col"] for c in inside)
        cells.append({"col": c0, "row": r0, "col_span": max(c["col"] for c in inside) - c0 + 1,
                      "row_span": max(c["row"] for c in inside) - r0 + 1, "box": [int(v) for v in span], "contents": None})
    return sorted(cells, key=lambda c: (c["row"], c["col"]))


def table_structure(detections, size_hw, offset_xy, categories=TABLE_CATEGORIES):
    """One table crop's detections -> the table record (table_structure_recognizer.py:210-257); boxes move from crop to
    page coordinates by the crop's offset."""
    h, w = size_hw
    ox, oy = offset_xy
    groups = {c: [] for c in categories}
    for box, score, label in zip(detections["boxes"], detections["scores"], detections["labels"]):
        b = [int(v) for v in box]
        groups[categories[int(label)]].append({"box": [b[0] + ox, b[1] + oy, b[2] + ox, b[3] + oy], "score": float(score)})
    groups = drop_nested_within_category(groups)
    row_boxes = sorted((e["box"] for e in groups["row"]), key=lambda b: b[1])
    col_boxes = sorted((e["box"] for e in groups["col"]), key=lambda b: b[0])
    cells = merge_span_cells(grid_cells(row_boxes, col_boxes), [e["box"] for e in groups["span"]])
    rows = sorted(groups["row"], key=lambda e: e["box"][1])
    cols = sorted(groups["col"], key=lambda e: e["box"][0])
    spans = sorted(groups["span"], key=lambda e: e["box"][1])
    return {"box": [ox, oy, ox + w, oy + h], "n_row": len(rows), "n_col": len(cols), "rows": rows, "cols": cols, "spans": spans,
            "cells": cells, "order": 0}


# ------------------------------------------------------------------------------------------ reading_order.py
def _blocked(boxes, a, b, axis):
    """Another box lies wholly in the gap between a and b along `axis` (1: stacked vertically, judged among boxes that share
    columns with a; 0: side by side, among boxes that share rows with a) - reading_order.py:83-121."""
    lo, hi = axis, axis + 2
    for k, s in enumerate(boxes):
        if k == a or k == b:
            continue
        if not (columns_overlap(s, boxes[a]) if axis == 1 else rows_overlap(s, boxes[a])):
            continue
        if boxes[a][hi] < s[lo] < boxes[b][lo] and boxes[a][hi] < s[hi] < boxes[b][lo]:
            return True
        if boxes[b][hi] < s[lo] < boxes[a][lo] and boxes[b][hi] < s[hi] < boxes[a][lo]:
            return True
    return False


def _graph(boxes, direction):
    """children[i] (ordered), parents[i] (in insertion order), distance[i] - reading_order.py:124-198.  Every ordered
    pair (i, j) is visited, as the reference's double loop does: the insertion order of links decides ties later on."""
    n = len(boxes)
    children = [[] for _ in range(n)]
    parents = [[] for _ in range(n)]

    def link(a, b):
        if b not in children[a]:
            children[a].append(b)
            parents[b].append(a)

    max_x = max(b[2] for b in boxes)
    distance = [0] * n
    for i in range(n):
        for j in range(n):
            if i == j:
                continue
            if direction == "top2bottom":
                if columns_overlap(boxes[i], boxes[j]) and not _blocked(boxes, i, j, 1):
                    if boxes[i][1] < boxes[j][1]:
                        link(i, j)
                    else:
                        link(j, i)
            elif rows_overlap(boxes[i], boxes[j]) and not _blocked(boxes, i, j, 0):
                ti, tj = boxes[i][2], boxes[j][2]
                if direction == "right2left":
                    if ti < tj:
                        link(j, i)
                    else:
                        link(i, j)
                else:  # left2right
                    if tj < ti:
                        link(j, i)
                    else:
                        link(i, j)
        if direction == "top2bottom":
            distance[i] = boxes[i][0] + boxes[i][1]
        elif direction == "right2left":
            distance[i] = (max_x - boxes[i][2]) + boxes[i][1]
        else:
            distance[i] = boxes[i][0] * 1 + boxes[i][1] * 5
    key = 0 if direction == "top2bottom" else 1
    for i in range(n):
        children[i] = sorted(children[i], key=lambda c: boxes[c][key])
    return children, parents, distance


def _walk(boxes, children, parents, distance, direction):
    """_priority_dfs (reading_order.py:14-80), on indices.  `children` is consumed as the walk goes, as in the reference."""
    n = len(boxes)
    if n == 0:
        return []
    pending = sorted(range(n), key=lambda i: distance[i])
    seen = [False] * n
    stack = [pending.pop(0)]
    order, waiting = [], []
    while not all(seen):
        while stack:
            progressed = False
            cur = stack.pop()
            if not seen[cur]:
                if all(seen[p] for p in parents[cur]) or len(parents[cur]) == 0:
                    seen[cur] = True
                    order.append(cur)
                    progressed = True
                elif cur not in waiting:
                    waiting.append(cur)
            if progressed:
                for w in list(reversed(waiting)):
                    stack.append(w)
                    waiting.remove(w)
            if len(children[cur]) > 0:
                stack.append(cur)
            if len(children[cur]) == 0:
                kids = []
                # (the reference removes from the list it is iterating: an element right behind a removed one is skipped)
                idx = 0
                while idx < len(stack):
                    node = stack[idx]
                    if cur in parents[node]:
                        kids.append(node)
                        stack.remove(node)
                    idx += 1
                if direction in "top2bottom":
                    kids = sorted(kids, key=lambda c: boxes[c][0], reverse=True)
                elif direction in ("right2left", "left2right"):
                    kids = sorted(kids, key=lambda c: boxes[c][1], reverse=True)
                stack.extend(kids)
                continue
            stack.append(children[cur].pop(0))
        picked = False
        for node in pending:
            if node in waiting:
                continue
            stack.append(node)
            pending.remove(node)
            picked = True
            break
        if not picked and not all(seen) and len(waiting) != 0:
            node = waiting.pop(0)
            seen[node] = True
            order.append(node)
    return order


def reading_order(elements, direction):
    """Writes element["order"] for every element of the list (fewer than two: left alone) and returns the list."""
    if len(elements) < 2:
        return elements
    if direction not in ("top2bottom", "right2left", "left2right"):
        raise ValueError(f"Invalid direction: {direction}")
    boxes = [e["box"] for e in elements]
    children, parents, distance = _graph(boxes, direction)
    for rank, index in enumerate(_walk(boxes, children, parents, distance, direction)):
        elements[index]["order"] = rank
    return elements


# ------------------------------------------------------------------------------------------ document_analyzer.py:19-217
_HIRAGANA = re.compile("^[぀-ゟ]+$")
_KATAKANA = re.compile("^[゠-ヿ]+$")


def dominant_direction(paragraphs):
    area = {"horizontal": 0, "vertical": 0}
    for p in paragraphs:
        x1, y1, x2, y2 = p["box"]
        area["horizontal" if p["direction"] == "horizontal" else "vertical"] += (x2 - x1) * (y2 - y1)
    return "vertical" if area["vertical"] > area["horizontal"] else "horizontal"


def _robust_threshold(sizes):
    ordered = sorted(sizes)
    median = ordered[len(ordered) // 2]
    if median == 0:
        return None
    spread = sorted(abs(s - median) for s in sizes)[len(ordered) // 2]
    if spread == 0:
        return None
    t = median - 2 * spread
    return t if t > 0 else None


def ruby_size_threshold(sizes, k):
    n = len(sizes)
    if n < 3:
        return None
    logs = [math.log(s) for s in sizes]
    bins = max(8, int(math.sqrt(n)))
    lo_v, hi_v = min(logs), max(logs)
    if hi_v - lo_v < 1e-9:
        return None
    width = (hi_v - lo_v) / bins
    hist = [0] * bins
    for v in logs:
        hist[min(int((v - lo_v) / width), bins - 1)] += 1
    first = max(range(bins), key=lambda i: hist[i])
    second, best = None, -1
    for i in range(bins):
        if abs(i - first) >= 2 and hist[i] > best:
            second, best = i, hist[i]
    if second is None:
        return _robust_threshold(sizes)
    lo, hi = min(first, second), max(first, second)
    if hi - lo <= 1:
        return _robust_threshold(sizes)
    floor = min(hist[i] for i in range(lo + 1, hi))
    valleys = [i for i in range(lo + 1, hi) if hist[i] == floor]
    valley = valleys[len(valleys) // 2]
    if (hist[first] + hist[second]) / (2 * floor + 1e-6) >= k:
        return math.exp(lo_v + (valley + 0.5) * width)
    return _robust_threshold(sizes)


def without_ruby(words, k):
    if len(words) <= 1:
        return words
    sizes = [math.sqrt((w["box"][2] - w["box"][0]) * (w["box"][3] - w["box"][1])) for w in words]
    usable = [s for s in sizes if s > 0]
    if len(usable) < 2:
        return words
    t = ruby_size_threshold(usable, k)
    if t is None:
        return words
    kept = []
    for w, s in zip(words, sizes):
        if 0 < s < t:
            text = w["contents"].replace(" ", "")
            if _HIRAGANA.match(text) or _KATAKANA.match(text):
                continue
        kept.append(w)
    return kept


def words_inside(words, element, ignore_ruby=False, ruby_threshold=2.0):
    """-> (text or None, direction or None, membership flags): the words at least half inside the element, in reading order."""
    flags = [False] * len(words)
    inside = []
    for i, w in enumerate(words):
        box = quad_box(w["points"])
        if contains(element["box"], box, threshold=0.5):
            flags[i] = True
            inside.append({"box": box, "contents": w["content"], "direction": w["direction"], "order": 0, "role": None})
    if not inside:
        return None, None, flags
    directions = [w["direction"] for w in inside]
    direction = "horizontal" if directions.count("horizontal") > directions.count("vertical") else "vertical"
    if ignore_ruby:
        inside = without_ruby(inside, ruby_threshold)
        if not inside:
            return None, None, flags
    reading_order(inside, "left2right" if direction == "horizontal" else "right2left")
    return "\n".join(w["contents"] for w in sorted(inside, key=lambda w: w["order"])), direction, flags


def figures_with_their_paragraphs(paragraphs, figures):
    out = []
    taken = [False] * len(paragraphs)
    for fig in figures:
        inside = []
        for i, p in enumerate(paragraphs):
            if contains(fig["box"], p["box"], threshold=0.7):
                inside.append(p)
                taken[i] = True
        direction = dominant_direction(inside)
        reading_order(inside, "left2right" if direction == "horizontal" else "right2left")
        out.append({"box": list(fig["box"]), "order": 0, "paragraphs": sorted(inside, key=lambda p: p["order"]), "direction": direction,
                    "figure_path": None})
    return out, taken


def aggregate(words, layout, ignore_meta=False, reading_order_opt="auto", ignore_ruby=False, ruby_threshold=2.0):
    """words: [{"points", "content", "direction", "rec_score", "det_score"}]; layout: {"paragraphs", "tables", "figures"} ->
    the page record {"paragraphs", "tables", "figures", "words"} (document_analyzer.py:487-601).  Cells of `layout`'s tables
    receive their contents in place."""
    used = [False] * len(words)
    for table in layout["tables"]:
        for cell in table["cells"]:
            text, _, flags = words_inside(words, cell, ignore_ruby, ruby_threshold)
            cell["contents"] = "" if text is None else text
            used = [a or b for a, b in zip(used, flags)]
    paragraphs = []
    for element in layout["paragraphs"]:
        text, direction, flags = words_inside(words, element, ignore_ruby, ruby_threshold)
        if text is None:
            continue
        used = [a or b for a, b in zip(used, flags)]
        paragraphs.append({"box": list(element["box"]), "contents": text, "direction": direction, "order": 0, "role": element["role"]})
    for w, u in zip(words, used):
        if not u:
            paragraphs.append({"box": quad_box(w["points"]), "contents": w["content"], "direction": w["direction"], "order": 0, "role": None})
    figures, in_figure = figures_with_their_paragraphs(paragraphs, layout["figures"])
    paragraphs = [p for p, f in zip(paragraphs, in_figure) if not f]
    page_direction = dominant_direction(paragraphs)
    headers = [p for p in paragraphs if p["role"] == "page_header" and not ignore_meta]
    footers = [p for p in paragraphs if p["role"] == "page_footer" and not ignore_meta]
    body = [p for p in paragraphs if p["role"] is None or p["role"] == "section_headings"]
    elements = body + layout["tables"] + figures
    reading_order(headers, "left2right")
    reading_order(footers, "left2right")
    if reading_order_opt == "auto":
        direction = "right2left" if page_direction == "vertical" else "top2bottom"
    else:
        direction = reading_order_opt
    reading_order(elements, direction)
    for e in elements:
        e["order"] += len(headers)
    for f in footers:
        f["order"] += len(elements) + len(headers)
    return {"paragraphs": sorted(headers + body + footers, key=lambda p: p["order"]),
            "tables": sorted(layout["tables"], key=lambda t: t["order"]),
            "words": words,
            "figures": sorted(figures, key=lambda f: f["order"])}


# ------------------------------------------------------------------------------------------ document_analyzer.py:239-423
def _edge(p, q):
    return math.sqrt((p[0] - q[0]) ** 2 + (p[1] - q[1]) ** 2)


def quad_is_vertical(quad, aspect=2):
    return _edge(quad[1], quad[2]) > _edge(quad[0], quad[1]) * aspect


def quad_is_noise(quad, thresh=15):
    return _edge(quad[0], quad[1]) < thresh or _edge(quad[1], quad[2]) < thresh


def split_text_across_cells(points, scores, tables):
    """The detector's quads cut at the cell borders of the tables they lie in (split_text_across_cells=True): a word at least half
    inside a table goes to the row (horizontal text) or column (vertical text: taller than twice its width) it overlaps most -
    the first one on ties - and is cut to every cell of that line it touches; pieces shorter than 15 px on either side are noise.
    Table words come first, table by table (horizontal pieces, then vertical ones), the untouched words after them in their
    order.  -> (points, scores)"""
    taken = [False] * len(points)
    out_p, out_s = [], []
    for table in tables:
        lying, standing = [], []
        for i, (quad, score) in enumerate(zip(points, scores)):
            if contains(table["box"], quad_box(quad), threshold=0.5):
                (standing if quad_is_vertical(quad) else lying).append((quad, score))
                taken[i] = True
        for words, lines, vertical in ((lying, table["rows"], False), (standing, table["cols"], True)):
            for quad, score in words:
                box = quad_box(quad)
                shares = [overlap_of_b(line["box"], box)[0] for line in lines]
                line_no = shares.index(max(shares)) + 1
                for cell in table["cells"]:
                    first, span = (cell["col"], cell["col_span"]) if vertical else (cell["row"], cell["row_span"])
                    if not (first <= line_no < first + span):
                        continue
                    inter = overlap_of_b(cell["box"], box)[1]
                    if inter is None:
                        continue
                    x1, y1, x2, y2 = inter
                    if vertical:
                        piece = [[quad[0][0], max(quad[0][1], y1)], [quad[1][0], max(quad[1][1], y1)],
                                 [quad[2][0], min(quad[2][1], y2)], [quad[3][0], min(quad[3][1], y2)]]
                    else:
                        piece = [[max(quad[0][0], x1), quad[0][1]], [min(quad[1][0], x2), quad[1][1]],
                                 [min(quad[2][0], x2), quad[2][1]], [max(quad[3][0], x1), quad[3][1]]]
                    if not quad_is_noise(piece):
                        out_p.append(piece)
                        out_s.append(score)
    for i, used in enumerate(taken):
        if not used:
            out_p.append(points[i])
            out_s.append(scores[i])
    return out_p, out_s

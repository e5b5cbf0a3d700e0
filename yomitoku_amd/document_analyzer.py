"""Page orchestrators: OCR, LayoutAnalyzer, DocumentAnalyzer (reference ocr.py:6-63,
layout_analyzer.py:7-49, document_analyzer.py:23-678): same constructors, nested `configs` keys and
return tuples.  The aggregation is integer box logic on the host and must give bit-exact word ->
cell / paragraph assignment and reading order.

MI355X specifics: the uint8 page is uploaded to HBM ONCE and shared by the four modules; the
det -> rec chain and the layout -> table chain still run concurrently (two Python threads, each
with its own HIP stream), as in document_analyzer.py:622-669.
"""

from __future__ import annotations

import os

import math
import re
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import imaging
from .geometry import calc_overlap_ratio, containment_matrix, is_contained, quad_to_xyxy
from .layout_parser import LayoutParser
from .reading_order import prediction_reading_order
from .schemas import (
    DocumentAnalyzerSchema,
    FigureSchema,
    LayoutAnalyzerSchema,
    OCRSchema,
    ParagraphSchema,
)
from .table_structure_recognizer import TableStructureRecognizer
from .text_detector import TextDetector
from .text_recognizer import TextRecognizer

_USAGE = "configs must be a dict. See the https://kotaro-kinoshita.github.io/yomitoku/module/#config"


# ---------------------------------------------------------------------------------------------- OCR
def ocr_aggregate(det_outputs, rec_outputs):
    return [
        {"points": p, "content": c, "direction": d, "det_score": ds, "rec_score": rs}
        for p, ds, c, rs, d in zip(det_outputs.points, det_outputs.scores, rec_outputs.contents, rec_outputs.scores,
                                   rec_outputs.directions)
    ]


class OCR:
    def __init__(self, configs={}, device="cuda", visualize=False):
        det_kwargs = {"device": device, "visualize": visualize}
        rec_kwargs = {"device": device, "visualize": visualize}
        if not isinstance(configs, dict):
            raise ValueError(_USAGE)
        det_kwargs.update(configs.get("text_detector", {}))
        rec_kwargs.update(configs.get("text_recognizer", {}))
        self.detector = TextDetector(**det_kwargs)
        self.recognizer = TextRecognizer(**rec_kwargs)

    def __call__(self, img):
        page = imaging.page_to_device(img, self.detector.device) if not isinstance(img, torch.Tensor) else img
        det_outputs, vis = self.detector(page)
        rec_outputs, vis = self.recognizer(page, det_outputs.points, vis=vis)
        return OCRSchema(words=ocr_aggregate(det_outputs, rec_outputs)), vis


# ---------------------------------------------------------------------------------------------- layout
class LayoutAnalyzer:
    def __init__(self, configs={}, device="cuda", visualize=False):
        lp_kwargs = {"device": device, "visualize": visualize}
        ts_kwargs = {"device": device, "visualize": visualize}
        if not isinstance(configs, dict):
            raise ValueError(_USAGE)
        lp_kwargs.update(configs.get("layout_parser", {}))
        ts_kwargs.update(configs.get("table_structure_recognizer", {}))
        self.layout_parser = LayoutParser(**lp_kwargs)
        self.table_structure_recognizer = TableStructureRecognizer(**ts_kwargs)

    def __call__(self, img):
        page = imaging.page_to_device(img, self.layout_parser.device) if not isinstance(img, torch.Tensor) else img
        layout_results, vis = self.layout_parser(page)
        table_boxes = [t.box for t in layout_results.tables]
        table_results, vis = self.table_structure_recognizer(page, table_boxes, vis=vis)
        results = LayoutAnalyzerSchema(paragraphs=layout_results.paragraphs, tables=table_results,
                                       figures=layout_results.figures)
        return results, vis


# ---------------------------------------------------------------------------------------------- aggregation helpers
def combine_flags(flag1, flag2):
    return [a or b for a, b in zip(flag1, flag2)]


def judge_page_direction(paragraphs):
    """'vertical' when vertical paragraphs cover more area than horizontal ones (:23-40)."""
    area = {"h": 0, "v": 0}
    for p in paragraphs:
        x1, y1, x2, y2 = p.box
        area["h" if p.direction == "horizontal" else "v"] += (x2 - x1) * (y2 - y1)
    return "vertical" if area["v"] > area["h"] else "horizontal"


def extract_paragraph_within_figure(paragraphs, figures):
    """Paragraphs at least 70 % inside a figure move into it, ordered inside the figure (:43-68)."""
    new_figures = []
    taken = [False] * len(paragraphs)
    for figure in figures:
        inside = []
        for i, paragraph in enumerate(paragraphs):
            if is_contained(figure.box, paragraph.box, threshold=0.7):
                inside.append(paragraph)
                taken[i] = True
        direction = judge_page_direction(inside)
        ordered = prediction_reading_order(inside, "left2right" if direction == "horizontal" else "right2left")
        new_figures.append(FigureSchema(box=figure.box, order=0, direction=direction,
                                        paragraphs=sorted(ordered, key=lambda p: p.order)))
    return new_figures, taken


_HIRAGANA = re.compile("^[\u3040-\u309F]+$")
_KATAKANA = re.compile("^[\u30A0-\u30FF]+$")


def _mad_threshold(sizes):
    ordered = sorted(sizes)
    n = len(ordered)
    median = ordered[n // 2]
    if median == 0:
        return None
    mad = sorted(abs(s - median) for s in sizes)[n // 2]
    if mad == 0:
        return None
    t = median - 2 * mad
    return t if t > 0 else None


def _compute_ruby_threshold(sizes, k):
    """Valley between the two dominant peaks of the log-size histogram when the bimodality is strong
    enough (sep >= k), else a median - 2 MAD fallback (:87-137)."""
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
    p1 = max(range(bins), key=lambda i: hist[i])
    p2, p2_val = None, -1
    for i in range(bins):
        if abs(i - p1) >= 2 and hist[i] > p2_val:
            p2, p2_val = i, hist[i]
    if p2 is None:
        return _mad_threshold(sizes)
    lo, hi = min(p1, p2), max(p1, p2)
    if hi - lo <= 1:
        return _mad_threshold(sizes)
    valley_val = min(hist[i] for i in range(lo + 1, hi))
    candidates = [i for i in range(lo + 1, hi) if hist[i] == valley_val]
    valley = candidates[len(candidates) // 2]
    if (hist[p1] + hist[p2]) / (2 * valley_val + 1e-6) >= k:
        return math.exp(lo_v + (valley + 0.5) * width)
    return _mad_threshold(sizes)


def filter_ruby(contained_words, element_direction, ruby_threshold):
    """Drop small kana-only words (furigana) below the size threshold (:140-172)."""
    if len(contained_words) <= 1:
        return contained_words
    sizes = [math.sqrt((w.box[2] - w.box[0]) * (w.box[3] - w.box[1])) for w in contained_words]
    valid = [s for s in sizes if s > 0]
    if len(valid) < 2:
        return contained_words
    threshold = _compute_ruby_threshold(valid, ruby_threshold)
    if threshold is None:
        return contained_words
    kept = []
    for word, s in zip(contained_words, sizes):
        if 0 < s < threshold:
            text = word.contents.replace(" ", "")
            if _HIRAGANA.match(text) or _KATAKANA.match(text):
                continue
        kept.append(word)
    return kept


def extract_words_within_element(pred_words, element, ignore_ruby=False, ruby_threshold=2.0):
    """Words at least 50 % inside `element`, joined in reading order (:175-217).
    Returns (text or None, direction or None, per-word membership flags)."""
    word_boxes = [quad_to_xyxy(word.points) for word in pred_words]
    flags = containment_matrix([element.box], word_boxes, 0.5)[0].tolist()
    return _words_of_element(pred_words, word_boxes, flags, ignore_ruby, ruby_threshold)


def _words_of_element(pred_words, word_boxes, flags, ignore_ruby, ruby_threshold):
    """Second half of extract_words_within_element, given the element's membership flags (aggregate()
    computes them for all cells and paragraphs of the page in one containment_matrix call)."""
    inside = [ParagraphSchema(box=word_boxes[i], contents=pred_words[i].content, direction=pred_words[i].direction,
                              order=0, role=None) for i in np.flatnonzero(flags).tolist()]
    if not inside:
        return None, None, flags
    dirs = [w.direction for w in inside]
    element_direction = "horizontal" if dirs.count("horizontal") > dirs.count("vertical") else "vertical"
    if ignore_ruby:
        inside = filter_ruby(inside, element_direction, ruby_threshold)
        if not inside:
            return None, None, flags
    prediction_reading_order(inside, "left2right" if element_direction == "horizontal" else "right2left")
    text = "\n".join(w.contents for w in sorted(inside, key=lambda w: w.order))
    return text, element_direction, flags


def is_vertical(quad, thresh_aspect=2):
    q = np.array(quad)
    return np.linalg.norm(q[1] - q[2]) > np.linalg.norm(q[0] - q[1]) * thresh_aspect


def is_noise(quad, thresh=15):
    q = np.array(quad)
    return np.linalg.norm(q[0] - q[1]) < thresh or np.linalg.norm(q[1] - q[2]) < thresh


def recursive_update(original, new_data):
    for key, value in new_data.items():
        if isinstance(value, dict) and isinstance(original.get(key), dict):
            recursive_update(original[key], value)
        else:
            original[key] = value
    return original


# ---- split_text_across_cells (:283-423)
def _extract_words_within_table(words, table, check_list):
    horizontal, vertical = [], []
    for i, (points, score) in enumerate(zip(words.points, words.scores)):
        if is_contained(table.box, quad_to_xyxy(points), threshold=0.5):
            (vertical if is_vertical(points) else horizontal).append({"points": points, "score": score})
            check_list[i] = True
    return horizontal, vertical, check_list


def _calc_overlap_words_on_lines(lines, words):
    return [[calc_overlap_ratio(line.box, quad_to_xyxy(word["points"]))[0] for line in lines] for word in words]


def _clip_words_to_cells(overlaps, table, words, vertical):
    """Cut each word at the borders of the cells of its best-matching column (vertical text) or row."""
    new_points, new_scores = [], []
    for word, ratios in zip(words, overlaps):
        line = ratios.index(max(ratios)) + 1
        pts, score = word["points"], word["score"]
        for cell in table.cells:
            start, span = (cell.col, cell.col_span) if vertical else (cell.row, cell.row_span)
            if not (start <= line < start + span):
                continue
            _, inter = calc_overlap_ratio(cell.box, quad_to_xyxy(pts))
            if inter is None:
                continue
            x1, y1, x2, y2 = inter
            if vertical:
                cut = [[pts[0][0], max(pts[0][1], y1)], [pts[1][0], max(pts[1][1], y1)],
                       [pts[2][0], min(pts[2][1], y2)], [pts[3][0], min(pts[3][1], y2)]]
            else:
                cut = [[max(pts[0][0], x1), pts[0][1]], [min(pts[1][0], x2), pts[1][1]],
                       [min(pts[2][0], x2), pts[2][1]], [max(pts[3][0], x1), pts[3][1]]]
            if not is_noise(cut):
                new_points.append(cut)
                new_scores.append(score)
    return new_points, new_scores


def _correct_vertical_word_boxes(overlap_ratios_vertical, table, table_words_vertical):
    return _clip_words_to_cells(overlap_ratios_vertical, table, table_words_vertical, vertical=True)


def _correct_horizontal_word_boxes(overlap_ratios_horizontal, table, table_words_horizontal):
    return _clip_words_to_cells(overlap_ratios_horizontal, table, table_words_horizontal, vertical=False)


def _split_text_across_cells(results_det, results_layout):
    check_list = [False] * len(results_det.points)
    new_points, new_scores = [], []
    for table in results_layout.tables:
        horizontal, vertical, check_list = _extract_words_within_table(results_det, table, check_list)
        ph, sh = _correct_horizontal_word_boxes(_calc_overlap_words_on_lines(table.rows, horizontal), table, horizontal)
        pv, sv = _correct_vertical_word_boxes(_calc_overlap_words_on_lines(table.cols, vertical), table, vertical)
        new_points += ph + pv
        new_scores += sh + sv
    for i, used in enumerate(check_list):
        if not used:
            new_points.append(results_det.points[i])
            new_scores.append(results_det.scores[i])
    results_det.points = new_points
    results_det.scores = new_scores
    return results_det


# ---------------------------------------------------------------------------------------------- DocumentAnalyzer
def _chain_priorities():
    spec = os.environ.get("YMK_CHAIN_PRIORITY", "")  # e.g. "layout:-1,ocr:0" (measurement knob)
    out = {}
    for item in filter(None, spec.split(",")):
        name, _, value = item.partition(":")
        out[name.strip()] = int(value)
    return out


class DocumentAnalyzer:
    def __init__(self, configs={}, device="cuda", visualize=False, ignore_meta=False, reading_order="auto",
                 split_text_across_cells=False, ignore_ruby=False, ruby_threshold=2.0):
        default_configs = {
            "ocr": {
                "text_detector": {"device": device, "visualize": visualize},
                "text_recognizer": {"device": device, "visualize": visualize},
            },
            "layout_analyzer": {
                "layout_parser": {"device": device, "visualize": visualize},
                "table_structure_recognizer": {"device": device, "visualize": visualize},
            },
        }
        self.reading_order = reading_order
        if not isinstance(configs, dict):
            raise ValueError(_USAGE)
        recursive_update(default_configs, configs)
        self.text_detector = TextDetector(**default_configs["ocr"]["text_detector"])
        self.text_recognizer = TextRecognizer(**default_configs["ocr"]["text_recognizer"])
        self.layout = LayoutAnalyzer(configs=default_configs["layout_analyzer"])
        self.visualize = visualize
        self.ignore_meta = ignore_meta
        self.split_text_across_cells = split_text_across_cells
        self.ignore_ruby = ignore_ruby
        self.ruby_threshold = ruby_threshold
        self._pool = ThreadPoolExecutor(max_workers=2)
        self._streams = {}
        # False: the two chains run one after the other on the caller's thread and stream (profiling: a kernel's
        # timing then brackets that kernel alone); results are the same either way
        self.concurrent_chains = True
        # HIP stream priority per chain (0 = default, -1 = high).  Measured on the analyzer bench: raising either chain
        # LOWERS the throughput (66.9 -> 62.3 layout high, 64.2 ocr high), so the default is no priority
        self.chain_priority = _chain_priorities()
        # `handover`: an object with any of maps / boxes / layout_raw / table_boxes / layouts (yomitoku_amd.testing.Handover) that
        # sees - and may replace - what one stage of the multi-page paths hands to the next, AFTER the stage has produced it at
        # full cost.  None (always, outside measurements): the stages hand over what they computed.  It exists so that a
        # throughput measurement with seeded weights (whose detections are noise) can feed realistic unit counts downstream
        # without a subclass that re-states the stage bodies (bench.py; tests/test_serving_gpu.py holds the hook to
        # "identity in, identical results out")
        self.handover = None

    # ---- aggregation (:487-601)
    def aggregate(self, ocr_res, layout_res, img=None):
        """`img` (the page, for its size): defaults to `self.img` as in the reference; the multi-page paths pass it so
        that no page's aggregation depends on shared state."""
        if img is None:
            img = self.img
        paragraphs = []
        cells = [cell for table in layout_res.tables for cell in table.cells]
        word_boxes = [quad_to_xyxy(word.points) for word in ocr_res.words]
        # word -> cell / paragraph membership for the whole page at once (same flags as the per-element form)
        member = containment_matrix([e.box for e in cells] + [p.box for p in layout_res.paragraphs], word_boxes, 0.5)
        used = np.zeros(len(ocr_res.words), dtype=bool)
        for cell, flags in zip(cells, member):
            words, _, _ = _words_of_element(ocr_res.words, word_boxes, flags, self.ignore_ruby, self.ruby_threshold)
            cell.contents = "" if words is None else words
            used |= flags
        for paragraph, flags in zip(layout_res.paragraphs, member[len(cells):]):
            words, direction, _ = _words_of_element(ocr_res.words, word_boxes, flags, self.ignore_ruby, self.ruby_threshold)
            if words is None:
                continue
            used |= flags
            paragraphs.append(ParagraphSchema(contents=words, box=paragraph.box, direction=direction, order=0,
                                              role=paragraph.role))
        for word, taken in zip(ocr_res.words, used):
            if not taken:
                paragraphs.append(ParagraphSchema(contents=word.content, box=quad_to_xyxy(word.points),
                                                  direction=word.direction, order=0, role=None))
        figures, in_figure = extract_paragraph_within_figure(paragraphs, layout_res.figures)
        paragraphs = [p for p, f in zip(paragraphs, in_figure) if not f]
        page_direction = judge_page_direction(paragraphs)
        headers = [p for p in paragraphs if p.role == "page_header" and not self.ignore_meta]
        footers = [p for p in paragraphs if p.role == "page_footer" and not self.ignore_meta]
        page_contents = [p for p in paragraphs if p.role is None or p.role == "section_headings"]
        elements = page_contents + layout_res.tables + figures
        prediction_reading_order(headers, "left2right")
        prediction_reading_order(footers, "left2right")
        if self.reading_order == "auto":
            order = "right2left" if page_direction == "vertical" else "top2bottom"
        else:
            order = self.reading_order
        prediction_reading_order(elements, order, img)
        for element in elements:
            element.order += len(headers)
        for footer in footers:
            footer.order += len(elements) + len(headers)
        return {
            "paragraphs": sorted(headers + page_contents + footers, key=lambda p: p.order),
            "tables": sorted(layout_res.tables, key=lambda t: t.order),
            "figures": sorted(figures, key=lambda f: f.order),
            "words": ocr_res.words,
        }

    # ---- the two concurrent chains, each on its own HIP stream
    def _submit(self, name, fn, *args):
        """A chain as a future: on its own thread + HIP stream, or - concurrent_chains False - run right here."""
        if self.concurrent_chains:
            return self._pool.submit(self._on_stream, name, fn, *args)
        from concurrent.futures import Future

        f = Future()
        try:
            f.set_result(fn(*args))
        except BaseException as exc:  # noqa: BLE001 - delivered by f.result(), like a pool future
            f.set_exception(exc)
        return f

    def _on_stream(self, name, fn, *args):
        dev = self.text_detector.device
        stream = self._streams.get(name)
        if stream is None:
            stream = self._streams[name] = torch.cuda.Stream(device=dev, priority=self.chain_priority.get(name, 0))
        stream.wait_stream(torch.cuda.default_stream(dev))  # the page upload was issued there
        with torch.cuda.stream(stream):
            out = fn(*args)
            stream.synchronize()
        return out

    def _detect_and_recognize(self, page):
        results_det, _ = self.text_detector(page)
        results_rec, ocr = self.text_recognizer(page, results_det.points, None)
        return results_det, results_rec, ocr

    def run(self, page):
        if self.split_text_across_cells:
            f_det = self._submit("ocr", self.text_detector, page)
            f_lay = self._submit("layout", self.layout, page)
            results_det, _ = f_det.result()
            results_layout, layout = f_lay.result()
            results_det = _split_text_across_cells(results_det, results_layout)
            results_rec, ocr = self.text_recognizer(page, results_det.points, None)
        else:
            f_ocr = self._submit("ocr", self._detect_and_recognize, page)
            f_lay = self._submit("layout", self.layout, page)
            results_det, results_rec, ocr = f_ocr.result()
            results_layout, layout = f_lay.result()
        results_ocr = OCRSchema(words=ocr_aggregate(results_det, results_rec))
        outputs = self.aggregate(results_ocr, results_layout)
        return DocumentAnalyzerSchema(**outputs), ocr, layout

    # ---- several pages per call: device batches across pages (the per-page path above leaves an MI355X mostly idle:
    # batch-1 RT-DETR / PARSeq launches are grid-starved and every page pays its own greedy-decode loop).  The stages of
    # a wave (yomitoku_amd.serving.Wave) are separate methods: `analyze_pages` runs them as two concurrent chains,
    # `serve` as a pipeline with one thread and HIP stream per stage.
    def _handed(self, what, wave, value):
        fn = getattr(self.handover, what, None) if self.handover is not None else None
        return value if fn is None else fn(wave, value)

    def _stage_detect(self, wave):
        """DBNet forwards over same-size pages of the wave; the maps come back into the wave slot's pinned buffers."""
        wave.maps = self._handed("maps", wave, self.text_detector.forward_pages(wave.pages, ring=wave.ring))

    def _stage_boxes(self, wave):
        """DB box extraction (C++, GIL released), the pages of the wave concurrently."""
        wave.dets = self._handed("boxes", wave, self.text_detector.extract_boxes(wave.maps, wave.sizes))

    def _stage_split(self, wave):
        wave.dets = [_split_text_across_cells(d, l) for d, l in zip(wave.dets, wave.lays)]

    def _stage_crops(self, wave):
        """Per-page mini-batches (bucketing, width budget) and the crop kernels that build their tensors."""
        wave.rec_plan = self.text_recognizer.plan_pages(wave.pages, [d.points for d in wave.dets])

    def _stage_recognize(self, wave, lane=0):
        """One grouped PARSeq forward over the mini-batches of all pages of the wave, on the recogniser handle of `lane`
        (serve keeps two forwards in flight: TextRecognizer.replica_model); with `rec_orientation_fallback` the decode
        happens here too, because the retry runs further forwards."""
        self.text_recognizer.forward_plan(wave.rec_plan, self.text_recognizer.replica_model(lane))
        if self.text_recognizer.rec_orientation_fallback:
            self._stage_decode(wave)

    def _stage_decode(self, wave):
        """Token decode and un-permutation on the host."""
        if wave.recs is None:
            wave.recs = self.text_recognizer.finish_plan(wave.rec_plan)
            wave.rec_plan = None

    def _stage_layout(self, wave):
        """One RT-DETRv2 layout forward over the pages (device half)."""
        wave.lay_raw = self._handed("layout_raw", wave, self.layout.layout_parser.forward_pages(wave.pages))

    def _stage_tables(self, wave):
        """Layout boxes on the host, then one table-structure forward over all table crops of the wave."""
        wave.lay_parsed = self.layout.layout_parser.pages_from_raw(wave.lay_raw)
        wave.lay_raw = None
        boxes = self._handed("table_boxes", wave, [[t.box for t in l.tables] for l in wave.lay_parsed])
        wave.tab_raw = self.layout.table_structure_recognizer.forward_tables(wave.pages, boxes)

    def _stage_cells(self, wave):
        """Row / column / span filters and the cell grids on the host."""
        tables = self.layout.table_structure_recognizer.tables_from_raw(wave.tab_raw, len(wave.pages))
        wave.tab_raw = None
        lays = [LayoutAnalyzerSchema(paragraphs=l.paragraphs, tables=t, figures=l.figures) for l, t in zip(wave.lay_parsed, tables)]
        wave.lays = self._handed("layouts", wave, lays)

    def _stage_finish(self, wave, k):
        """Aggregation of page k of the wave -> DocumentAnalyzerSchema."""
        results_ocr = OCRSchema(words=ocr_aggregate(wave.dets[k], wave.recs[k]))
        return DocumentAnalyzerSchema(**self.aggregate(results_ocr, wave.lays[k], img=wave.imgs[k]))

    def _recognize_wave(self, wave):
        self._stage_crops(wave)
        self._stage_recognize(wave)
        self._stage_decode(wave)

    def _layout_wave(self, wave):
        self._stage_layout(wave)
        self._stage_tables(wave)
        self._stage_cells(wave)

    def _ocr_wave(self, wave):
        self._stage_detect(wave)
        self._stage_boxes(wave)
        if not self.split_text_across_cells:
            self._recognize_wave(wave)

    def analyze_pages(self, imgs, wave: int = 8):
        """`__call__` over a list of pages, `wave` pages at a time on the device.  Every page's result is what
        `__call__(img)` returns for it (pages never interact: batches are per-image independent, recogniser
        mini-batches are formed per page); what changes is how the launches are shared.  The two chains of a wave
        run concurrently on their own HIP streams, as in `run`.  Returns [(DocumentAnalyzerSchema, None, None), ...].
        One wave at a time: `serve` is the throughput path (several waves in flight, failures isolated per page)."""
        from .serving import Wave

        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        dev = self.text_detector.device
        out = []
        size = max(1, int(wave))
        for start in range(0, len(imgs), size):
            chunk = imgs[start : start + size]
            pages = [img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, dev) for img in chunk]
            w = Wave(start // size, range(start, start + len(chunk)), chunk, pages)
            f_ocr = self._submit("ocr", self._ocr_wave, w)
            f_lay = self._submit("layout", self._layout_wave, w)
            f_ocr.result()
            f_lay.result()
            if self.split_text_across_cells:
                self._stage_split(w)
                self._submit("ocr", self._recognize_wave, w).result()
            out.extend((self._stage_finish(w, k), None, None) for k in range(len(chunk)))
        return out

    def serve(self, sources, wave: int = 16, in_flight: int = 4, defer_full_gc: bool = True, with_source: bool = False, rec_lanes: int = 2):
        """The multi-page entry point: host pages (uint8 H x W x 3 BGR arrays) and / or image file paths in, one result
        per page out, in page order - the page loop of cli/main.py:105-137 as a stage pipeline on one GPU from one
        process (yomitoku_amd/serving.py): pinned staging + H2D on a copy stream, `wave` pages per device batch (16 measured best
        at 1600 x 1200: one greedy loop and forwards of equal size per wave; 8: 5 % less, 32: 3 % less), up to `in_flight` waves
        between upload and aggregation.  A page's entry is its DocumentAnalyzerSchema - equal to
        `__call__(img)[0]` - or, when the page (or its file) failed, the exception object; the other pages are not
        affected (cli/main.py:555-564).  `defer_full_gc`: postpone CPython's generation-2 garbage collections until
        the job is done (a full pass holds the GIL for 100+ ms with a few hundred results alive and stalls every stage).
        `with_source`: (source index, frame index, entry) triples, for callers that write one output per file page.
        `rec_lanes`: recogniser forwards in flight (2: a second PARSeq handle with the same weights; serving.py)."""
        from .serving import PagePipeline

        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        pipe = getattr(self, "_pipeline", None)
        if pipe is None or (pipe.wave, pipe.in_flight, pipe.rec_lanes_asked) != (max(1, int(wave)), max(1, int(in_flight)), int(rec_lanes)):
            if pipe is not None:
                pipe.close()
            pipe = self._pipeline = PagePipeline(self, wave=wave, in_flight=in_flight, rec_lanes=rec_lanes)
            pipe.rec_lanes_asked = int(rec_lanes)
        pipe.defer_full_gc = bool(defer_full_gc)
        return pipe.serve(sources, with_source=with_source)

    def close(self, release_models: bool = True):
        """Stop the pipeline threads, release the extra recogniser handles and - `release_models` - the device memory of the
        four nets (weights and the reserved workspaces: tens of GB per analyzer).  The analyzer stays usable: a net rebuilds
        its handle from the state dict it holds the next time it is called."""
        pipe = getattr(self, "_pipeline", None)
        if pipe is not None:
            pipe.close()
            self._pipeline = None
        self.text_recognizer.close_replicas()
        if release_models:
            for module in (self.text_detector, self.text_recognizer, self.layout.layout_parser, self.layout.table_structure_recognizer):
                module.model.close()

    def __call__(self, img):
        self.img = img
        dev = self.text_detector.device
        page = img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, dev)
        results, ocr, layout = self.run(page)
        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        return results, ocr, layout

"""Measurement hooks for the multi-page paths (`DocumentAnalyzer.handover`): NOT part of the reference's surface.

Seeded random weights detect noise, so a throughput measurement that wants a trained pipeline's unit counts (text lines,
table crops, cells per page) has to put ground truth into the DISCRETE hand-overs between stages while every network and
every kernel runs at full cost.  Until round 5 bench.py did that with a subclass that re-stated three stage bodies of
`DocumentAnalyzer` - a copy that could drift from the product.  Now the product's own stage bodies run, unmodified, and call
`analyzer.handover.<point>(wave, value)` with what they produced; the object below decides what travels on:

    maps          the detector's probability maps of the wave's pages (host arrays)            after the DBNet forwards + D2H
    boxes         [TextDetectorSchema] per page                                                after the C++ box extraction
    layout_raw    [(logits, boxes, (h, w))] per page                                           after the layout forward
    table_boxes   [[x1, y1, x2, y2] per table] per page                                        after the layout post-processing
    layouts       [LayoutAnalyzerSchema] per page                                              after the table filters / cell grids

`Handover` passes everything through (tests/test_serving_gpu.py: results identical to handover = None).  A `truth` entry
is any object with `quads`, `tables`, `paragraphs` (lists of boxes in page coordinates) and `truth_map` (the probability
map a trained detector would emit: bench.render_truth_map); wave page id i belongs to truth[i % len(truth)]."""

from __future__ import annotations

import numpy as np

from .schemas import Element, LayoutAnalyzerSchema, TextDetectorSchema


class Handover:
    """Identity at every point; subclasses replace what they need.  `stats`: optional dict of lists the hooks append what
    the product's own stages had produced to (unit counts of the discarded outputs)."""

    def __init__(self, truth=None, stats=None):
        self.truth = truth
        self.stats = stats

    def _truth(self, wave):
        return [self.truth[i % len(self.truth)] for i in wave.ids]

    def _note(self, key, values):
        if self.stats is not None:
            self.stats.setdefault(key, []).extend(values)

    def maps(self, wave, maps):
        return maps

    def boxes(self, wave, dets):
        return dets

    def layout_raw(self, wave, raw):
        return raw

    def table_boxes(self, wave, boxes):
        return boxes

    def layouts(self, wave, lays):
        return lays


class TruthHandover(Handover):
    """bench.py's headline workload: the box extraction runs on the map a trained detector would emit and the recogniser gets
    the page's true text lines; the table-structure net gets the true table boxes; the aggregation gets the true paragraphs
    with the PRODUCT'S table structures.  Every forward, the extraction, the layout post-processing and the table filters run
    at full cost on the way."""

    def maps(self, wave, maps):
        truth = self._truth(wave)
        assert all(m.shape == t.truth_map.shape for m, t in zip(maps, truth)), "rendered map and detector map differ in shape"
        return [t.truth_map for t in truth]

    def boxes(self, wave, dets):
        self._note("det_boxes", (len(d.points) for d in dets))
        return [TextDetectorSchema(points=t.quads, scores=[1.0] * len(t.quads)) for t in self._truth(wave)]

    def table_boxes(self, wave, boxes):
        self._note("layout_boxes", (len(l.paragraphs) + len(l.tables) + len(l.figures) for l in wave.lay_parsed))
        return [list(t.tables) for t in self._truth(wave)]

    def layouts(self, wave, lays):
        self._note("cells", (sum(len(x.cells) for x in l.tables) for l in lays))
        out = []
        for t, l in zip(self._truth(wave), lays):
            paragraphs = [Element(id=None, box=b, score=1.0, role=None, contents=None) for b in t.paragraphs]
            out.append(LayoutAnalyzerSchema(paragraphs=paragraphs, tables=l.tables, figures=[]))
        return out


class NetOutputHandover(Handover):
    """bench.py's control leg: only the two NETWORK OUTPUTS a trained checkpoint would produce are substituted - the detector's
    map and the layout net's raw (logits, boxes), one query per true paragraph / table at score 0.98 - and everything
    downstream is the product's own code on its own hand-overs."""

    def __init__(self, truth=None, stats=None, categories=None):
        super().__init__(truth, stats)
        self.categories = categories  # {category name: class index} of the layout parser

    def maps(self, wave, maps):
        truth = self._truth(wave)
        assert all(m.shape == t.truth_map.shape for m, t in zip(maps, truth)), "rendered map and detector map differ in shape"
        return [t.truth_map for t in truth]

    def layout_raw(self, wave, raw):
        out = []
        for (logits, boxes, (h, w)), t in zip(raw, self._truth(wave)):
            lg = np.full_like(logits, -12.0)
            bx = np.zeros_like(boxes)
            units = [(b, self.categories["paragraphs"]) for b in t.paragraphs] + [(b, self.categories["tables"]) for b in t.tables]
            for q, ((x0, y0, x1, y1), c) in enumerate(units[: lg.shape[1]]):
                lg[0, q, c] = 4.0
                bx[0, q] = ((x0 + x1) / 2 / w, (y0 + y1) / 2 / h, (x1 - x0) / w, (y1 - y0) / h)
            out.append((lg, bx, (h, w)))
        return out

"""TableStructureRecognizer module (reference table_structure_recognizer.py:21-290).

One difference in *how*: the reference runs the RT-DETRv2 once per table, serially, batch 1
(:261-278); here every table crop of the page goes through ONE batched forward (results are
per-image independent, tests/test_rtdetr_gpu.py::test_batch_of_pages_matches_oracle)."""

from __future__ import annotations

import torch

from . import imaging
from .base import BaseModelCatalog, BaseModule
from .configs import TableStructureRecognizerRTDETRv2Config
from .geometry import calc_intersection, filter_by_flag, is_contained
from .layout_parser import RTDETRPostProcessor, filter_contained_rectangles_within_category
from .nets import RTDETRv2
from .schemas import TableStructureRecognizerSchema


class TableStructureRecognizerModelCatalog(BaseModelCatalog):
    def __init__(self):
        super().__init__()
        self.register("rtdetrv2", TableStructureRecognizerRTDETRv2Config, RTDETRv2)


def extract_cells(row_boxes, col_boxes):
    """Cell grid = every non-empty row x column intersection, 1-based (row, col), row-major - :27-46.  All
    intersections at once (integer max / min of the truncated coordinates, as calc_intersection does per pair)."""
    if len(row_boxes) == 0 or len(col_boxes) == 0:
        return []
    import numpy as np

    r = np.trunc(np.asarray(row_boxes, dtype=np.float64).reshape(-1, 4)).astype(np.int64)
    c = np.trunc(np.asarray(col_boxes, dtype=np.float64).reshape(-1, 4)).astype(np.int64)
    x1 = np.maximum(r[:, None, 0], c[None, :, 0])
    y1 = np.maximum(r[:, None, 1], c[None, :, 1])
    x2 = np.minimum(r[:, None, 2], c[None, :, 2])
    y2 = np.minimum(r[:, None, 3], c[None, :, 3])
    ii, jj = np.nonzero((x2 - x1 > 0) & (y2 - y1 > 0))  # row-major: the order of the nested loops
    boxes = np.stack([x1[ii, jj], y1[ii, jj], x2[ii, jj], y2[ii, jj]], axis=1).tolist()
    return [{"col": j + 1, "row": i + 1, "col_span": 1, "row_span": 1, "box": box, "contents": None}
            for i, j, box in zip(ii.tolist(), jj.tolist(), boxes)]


def _extract_cells_scalar(row_boxes, col_boxes):
    """The nested loops of table_structure_recognizer.py:27-46 (tests compare the matrix form with them)."""
    cells = []
    for i, row_box in enumerate(row_boxes):
        for j, col_box in enumerate(col_boxes):
            inter = calc_intersection(row_box, col_box)
            if inter is not None:
                cells.append({"col": j + 1, "row": i + 1, "col_span": 1, "row_span": 1, "box": inter, "contents": None})
    return cells


def filter_contained_cells_within_spancell(cells, span_boxes):
    """Merge the grid cells inside each detected span into one spanning cell - :49-85."""
    keep = [True] * len(cells)
    members = [[] for _ in span_boxes]
    for i, span_box in enumerate(span_boxes):
        for j, cell in enumerate(cells):
            if is_contained(span_box, cell["box"]):
                keep[j] = False
                members[i].append(cell)
    cells = filter_by_flag(cells, keep)
    for span_box, group in zip(span_boxes, members):
        if not group:
            continue
        rows = [c["row"] for c in group]
        cols = [c["col"] for c in group]
        cells.append({"col": min(cols), "row": min(rows), "col_span": max(cols) - min(cols) + 1,
                      "row_span": max(rows) - min(rows) + 1, "box": [int(v) for v in span_box], "contents": None})
    return sorted(cells, key=lambda c: (c["row"], c["col"]))


class TableStructureRecognizer(BaseModule):
    model_catalog = TableStructureRecognizerModelCatalog()
    # bounds the activation workspace (0.6 GB of fp32 maps per table crop: 38 GB of 288 for 64).  64, not 16, since round 6: a page
    # full of tables brings ten crops, a wave of such pages 170 - three forwards instead of eleven, and the 20 x 20 level of the
    # net (400 rows per crop) fills the chip: one crop costs 0.71 ms at 64 per forward against 0.88 at 16
    # (profiles/r06_rtdetr_forward_alone_*.txt; the table-heavy serve leg 36 -> 44 pages/s with this alone)
    MAX_TABLES_PER_FORWARD = 64

    def __init__(self, model_name="rtdetrv2", path_cfg=None, device="cuda", visualize=False, from_pretrained=True,
                 infer_onnx=False):
        super().__init__()
        if infer_onnx:
            raise NotImplementedError("the ONNX backend is out of scope of the MI355X path (infer_onnx=False only)")
        self.load_model(model_name, path_cfg, from_pretrained=from_pretrained)
        self.device = device
        self.visualize = visualize
        self.model.eval()
        self.postprocessor = RTDETRPostProcessor(
            num_classes=self._cfg.RTDETRTransformerv2.num_classes,
            num_top_queries=self._cfg.RTDETRTransformerv2.num_queries,
        )
        self.thresh_score = self._cfg.thresh_score
        self.label_mapper = {i: c for i, c in enumerate(self._cfg.category)}
        self.infer_onnx = False
        self.model.to(self.device)

    def preprocess(self, img, boxes):
        """All table crops of the page as one N x 3 x 640 x 640 device tensor + per-crop metadata."""
        page = img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device)
        return imaging.rtdetr_batch_tensor([page], [(0, box) for box in boxes], tuple(self._cfg.data.img_size))  # one launch for the page's crops

    def postprocess(self, preds, data):
        h, w = data["size"]
        outputs = self.postprocessor(preds, (w, h), self.thresh_score)[0]
        ox, oy = data["offset"]
        category_elements = {c: [] for c in self.label_mapper.values()}
        for box, score, label in zip(outputs["boxes"], outputs["scores"], outputs["labels"]):
            b = box.astype(int).tolist()
            category_elements[self.label_mapper[int(label)]].append(
                {"box": [b[0] + ox, b[1] + oy, b[2] + ox, b[3] + oy], "score": float(score)}
            )
        category_elements = filter_contained_rectangles_within_category(category_elements)
        cells, rows, cols, spans = self.extract_cell_elements(category_elements)
        table = {"box": [ox, oy, ox + w, oy + h], "n_row": len(rows), "n_col": len(cols), "rows": rows, "cols": cols,
                 "spans": spans, "cells": cells, "order": 0}
        return TableStructureRecognizerSchema(**table)

    def extract_cell_elements(self, elements):
        row_boxes = sorted((e["box"] for e in elements["row"]), key=lambda b: b[1])
        col_boxes = sorted((e["box"] for e in elements["col"]), key=lambda b: b[0])
        span_boxes = [e["box"] for e in elements["span"]]
        cells = filter_contained_cells_within_spancell(extract_cells(row_boxes, col_boxes), span_boxes)
        rows = sorted(elements["row"], key=lambda e: e["box"][1])
        cols = sorted(elements["col"], key=lambda e: e["box"][0])
        spans = sorted(elements["span"], key=lambda e: e["box"][1])
        return cells, rows, cols, spans

    def forward_tables(self, imgs, boxes_list):
        """The device half of `recognize_pages`: the table crops of ALL pages fill shared forwards of
        MAX_TABLES_PER_FORWARD crops.  Returns [(page index, logits 1 x Q x C, boxes 1 x Q x 4, meta)] per table, host
        arrays, in page / box order - what `tables_from_raw` turns into TableStructureRecognizerSchemas on the host."""
        pages = [img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device) for img in imgs]
        oh, ow = self._cfg.data.img_size
        self.model.reserve_once(self.MAX_TABLES_PER_FORWARD, int(oh), int(ow), self.device)  # any table count: no reallocation later (once per live handle)
        flat = [(p, box) for p, boxes in enumerate(boxes_list) for box in boxes]
        raw = []
        # forwards of EQUAL size: 18 crops run as 9 + 9, not 16 + 2 - a batch of two leaves most of the chip idle
        n_fwd = max(1, -(-len(flat) // self.MAX_TABLES_PER_FORWARD))
        per = -(-len(flat) // n_fwd) if flat else 1
        for start in range(0, len(flat), per):
            chunk = flat[start : start + per]
            batch, metas = imaging.rtdetr_batch_tensor(pages, chunk, (oh, ow))  # every crop of the forward in one launch
            preds = self.model(batch)
            logits, bxs = imaging.to_host(preds["pred_logits"], preds["pred_boxes"])
            for k, ((p, _), data) in enumerate(zip(chunk, metas)):
                raw.append((p, logits[k : k + 1], bxs[k : k + 1], data))
        return raw

    def tables_from_raw(self, raw, n_pages):
        """The host half: post-processor, row / column / span filters and the cell grid per table; per page the list
        `__call__` returns (tables without rows or columns dropped)."""
        outputs = [[] for _ in range(n_pages)]
        for p, logits, bxs, data in raw:
            table = self.postprocess({"pred_logits": logits, "pred_boxes": bxs}, data)
            if table.n_row > 0 and table.n_col > 0:
                outputs[p].append(table)
        return outputs

    def recognize_pages(self, imgs, boxes_list):
        """`__call__` for several pages: shared forwards over the table crops of all pages, then the host grid logic."""
        return self.tables_from_raw(self.forward_tables(imgs, boxes_list), len(imgs))

    def __call__(self, img, table_boxes, vis=None):
        outputs = []
        for start in range(0, len(table_boxes), self.MAX_TABLES_PER_FORWARD):
            batch, metas = self.preprocess(img, table_boxes[start : start + self.MAX_TABLES_PER_FORWARD])
            preds = self.model(batch)
            for i, data in enumerate(metas):
                one = {"pred_logits": preds["pred_logits"][i : i + 1], "pred_boxes": preds["pred_boxes"][i : i + 1]}
                table = self.postprocess(one, data)
                if table.n_row > 0 and table.n_col > 0:
                    outputs.append(table)
        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        return outputs, vis

"""CellDetector module (reference table_cell_detector.py:195-524): the table cell detector of `yomitoku_table` -
RT-DETRv2 at 960 x 960 with 1500 queries and 6 classes (table, cell, header, empty, kv_item, grid) - with the same
constructor, catalog name, config and `__call__(img, tables) -> [TableDetectorSchema]` contract.

On the MI355X: the table crops are resized on the device (PIL-exact antialiased bilinear kernel), ALL tables of a call
share batched forwards (the reference runs them one by one, :503-514; images of a batch are independent), and the
post-processing's OpenCV part (find_holes_as_rects: rectangle / morphologyEx / floodFill / findContours) runs in the
C++ host code of libymk_hip.so (ymk_table_hole_rects).  The box logic is host Python, pinned against the reference's own
functions (tests/golden/cells.json)."""

from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import _lib, imaging
from .base import BaseModelCatalog, BaseModule, load_config
from .configs import TableCellParserRTDETRv2Config
from .geometry import adjacency_matrices, calc_iou, containment_matrix, filter_by_flag, is_contained
from .layout_parser import RTDETRPostProcessor, load_local_checkpoint
from .nets import RTDETRv2
from .schemas import CellSchema, RegionSchema, TableDetectorSchema


class TableParserModelCatalog(BaseModelCatalog):
    def __init__(self):
        super().__init__()
        self.register("rtdetrv2", TableCellParserRTDETRv2Config, RTDETRv2)


def filter_contained_rectangles_with_category(category_elements, ignore_categories=()):
    """Per category, of two boxes one of which lies (80 %) inside the other, keep the INNER one; when they contain each
    other the larger one stays (what table_cell_detector.py:41-75 does - its comment says the opposite; the code is
    what is mirrored.  Note the direction differs from the layout filter, which keeps the outer box)."""
    for category, elements in category_elements.items():
        if category in ignore_categories:
            continue
        boxes = [e["box"] for e in elements]
        keep = [True] * len(boxes)
        for i, box_i in enumerate(boxes):
            for j in range(i + 1, len(boxes)):
                box_j = boxes[j]
                ij = is_contained(box_i, box_j)
                ji = is_contained(box_j, box_i)
                if ij and ji:
                    area_i = (box_i[2] - box_i[0]) * (box_i[3] - box_i[1])
                    area_j = (box_j[2] - box_j[0]) * (box_j[3] - box_j[1])
                    if area_i > area_j:
                        keep[j] = False
                    else:
                        keep[i] = False
                elif ij:
                    keep[i] = False
                elif ji:
                    keep[j] = False
        category_elements[category] = filter_by_flag(elements, keep)
    return category_elements


def filter_contained_rectangles_across_categories(category_elements, source, target):
    """`target` boxes more than 80 % inside some `source` box go (table_cell_detector.py:100-113): one containment matrix
    over all (source, target) pairs instead of the pair loop."""
    inside = containment_matrix([e["box"] for e in category_elements[source]], [e["box"] for e in category_elements[target]], 0.8)
    category_elements[target] = filter_by_flag(category_elements[target], (~inside.any(axis=0)).tolist())
    return category_elements


def find_holes_as_rects(table_shape, cell_boxes, pad=2, close_ksize=5, min_area=300):
    """Rectangles around the parts of the table crop that no cell covers (table_cell_detector.py:116-143), by the C++
    host routine ymk_table_hole_rects."""
    h, w = int(table_shape[0]), int(table_shape[1])
    boxes = np.ascontiguousarray(np.asarray([[int(v) for v in b] for b in cell_boxes], dtype=np.int32).reshape(-1, 4))
    cap = 256
    while True:
        out = np.empty((cap, 4), dtype=np.int32)
        n = ctypes.c_int()
        status = _lib.load().ymk_table_hole_rects(h, w, boxes.ctypes.data, len(boxes), int(pad), int(close_ksize), int(min_area),
                                                  out.ctypes.data, cap, ctypes.byref(n))
        if status != 0 and cap < (1 << 16) and b"capacity" in (_lib.load().ymk_last_error() or b""):
            cap *= 8
            continue
        _lib.check(status, "ymk_table_hole_rects")
        return out[: n.value].tolist()


_ROLES = ("cell", "header", "empty")


def choose_role(role_counts):
    """Majority role; among equals the earlier of cell / header / empty - which is the reference's "ties go to cell, else
    the first candidate" (table_cell_detector.py:146-158) because `cell` comes first."""
    if not role_counts:
        return None
    return max(role_counts, key=lambda r: (role_counts[r], -list(role_counts).index(r)))


def calc_adjacent_holes_to_cells(holes, cells):
    """A hole becomes a cell when detected cells touch it on more than two of its four sides; its role is the majority role
    over every (side, neighbour) contact (table_cell_detector.py:161-192).  All hole x cell contacts come from four
    adjacency matrices (geometry.adjacency_matrices) instead of 4 x holes x cells scalar predicates."""
    if not holes:
        return []
    hole_boxes, cell_boxes = [h["box"] for h in holes], [c["box"] for c in cells]
    right, down = adjacency_matrices(hole_boxes, cell_boxes)   # the cell is the hole's right / lower neighbour
    left, up = (m.T for m in adjacency_matrices(cell_boxes, hole_boxes))  # the hole is the cell's right / lower neighbour
    sides = right.any(1).astype(int) + left.any(1) + down.any(1) + up.any(1)
    contacts = right.astype(np.int64) + left + down + up  # a cell counts once per side it touches
    role_of = np.array([_ROLES.index(c["role"]) for c in cells], dtype=np.int64).reshape(-1)
    counts = np.stack([(contacts * (role_of == k)[None, :]).sum(1) for k in range(len(_ROLES))], axis=1) if cells else np.zeros((len(holes), 3), np.int64)
    kept = []
    for hole, n_sides, row in zip(holes, sides.tolist(), counts):
        if n_sides > 2:
            hole["role"] = _ROLES[int(np.argmax(row))]  # first maximum: cell before header before empty
            kept.append(hole)
    return kept


class CellDetector(BaseModule):
    model_catalog = TableParserModelCatalog()
    MAX_TABLES_PER_FORWARD = 8  # 960 x 960 inputs: 120 x 120 x 512 fp32 maps per table

    def __init__(self, model_name="rtdetrv2", path_cfg=None, device="cuda", visualize=False, from_pretrained=True,
                 infer_onnx=False):
        super().__init__()
        if infer_onnx:
            raise NotImplementedError("the ONNX backend is out of scope of the MI355X path (infer_onnx=False only)")
        # a local rtdetrv2_pytorch training checkpoint takes precedence over the hub weights (:214-233)
        default_cfg, _ = self.model_catalog.get(model_name)
        peek = load_config(default_cfg, path_cfg)
        weights_path = getattr(peek, "weights_path", None)
        use_local = bool(weights_path) and os.path.exists(weights_path)
        self.load_model(model_name, path_cfg, from_pretrained=(from_pretrained and not use_local))
        if use_local:
            load_local_checkpoint(self.model, weights_path, getattr(self._cfg, "weights_key", "ema"))
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

    def preprocess(self, img, tables):
        """All table crops of the page as one N x 3 x 960 x 960 device tensor + per-crop metadata (:318-337)."""
        page = img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device)
        oh, ow = (int(v) for v in self._cfg.data.img_size)
        batch = torch.empty((len(tables), 3, oh, ow), dtype=torch.float32, device=page.device)
        metas = []
        for i, table in enumerate(tables):
            _, size, offset = imaging.rtdetr_tensor(page, table.box, (oh, ow), out=batch[i])
            metas.append({"size": size, "offset": offset})
        return batch, metas

    def is_fully_contained(self, box1, box2, threshold=0.9):
        return calc_iou(box1, box2) >= threshold

    def postprocess(self, preds, data, table_box):
        h, w = data["size"]
        outputs = self.postprocessor(preds, (w, h), self.thresh_score)[0]
        category_elements = {category: [] for category in self.label_mapper.values()}
        category_elements["hole"] = []
        for box, score, label in zip(outputs["boxes"], outputs["scores"], outputs["labels"]):
            category = self.label_mapper[int(label)]
            box = box.astype(int).tolist()
            # a detection that coincides with the whole crop is dropped, except for the region classes that may cover it
            if category not in ("grid", "kv_item") and self.is_fully_contained(box, [0, 0, w, h]):
                continue
            category_elements[category].append({"box": box, "score": float(score), "role": category})
        category_elements = filter_contained_rectangles_with_category(category_elements, ignore_categories=["kv_item", "grid"])
        category_elements = filter_contained_rectangles_across_categories(category_elements, source="cell", target="header")
        category_elements = filter_contained_rectangles_across_categories(category_elements, source="cell", target="empty")
        cell_boxes = category_elements["cell"] + category_elements["header"] + category_elements["empty"]
        for box in find_holes_as_rects(data["size"], [cell["box"] for cell in cell_boxes]):
            category_elements["hole"].append({"box": box, "score": 1.0, "role": "hole"})
        ox, oy = data["offset"]
        for cells in category_elements.values():
            for cell in cells:
                cell["box"][0] += ox
                cell["box"][1] += oy
                cell["box"][2] += ox
                cell["box"][3] += oy
        if len(category_elements["cell"] + category_elements["empty"] + category_elements["header"]) == 0:
            category_elements["cell"] = [{"box": table_box, "role": "cell"}]  # no cell found: the table is one cell
        cells = self.extract_cell_elements(category_elements)
        cells = self.remove_noise_cells(cells, min_width=10, min_height=10)
        kv_regions = [RegionSchema(id=None, box=e["box"], role="kv_item", score=e["score"]) for e in category_elements.get("kv_item", [])]
        grid_regions = [RegionSchema(id=None, box=e["box"], role="grid", score=e["score"]) for e in category_elements.get("grid", [])]
        return cells, kv_regions, grid_regions

    def remove_noise_cells(self, cells, min_width=30, min_height=30):
        return [c for c in cells if c.box[2] - c.box[0] > min_width and c.box[3] - c.box[1] > min_height]

    def extract_cell_elements(self, elements):
        elements["hole"] = calc_adjacent_holes_to_cells(elements["hole"], list(elements["cell"] + elements["header"] + elements["empty"]))
        cells = []
        for category, values in elements.items():
            if category in ("cell", "header", "empty", "group", "hole"):
                for value in values:
                    cells.append(CellSchema(id=f"c{len(cells)}", box=value["box"], role=value["role"], contents=None, row=None,
                                            col=None, row_span=None, col_span=None))
        return cells

    def __call__(self, img, tables):
        outputs = []
        for start in range(0, len(tables), self.MAX_TABLES_PER_FORWARD):
            chunk = tables[start : start + self.MAX_TABLES_PER_FORWARD]
            batch, metas = self.preprocess(img, chunk)
            preds = self.model(batch)
            logits = preds["pred_logits"].cpu().numpy()
            boxes = preds["pred_boxes"].cpu().numpy()
            for i, (data, table) in enumerate(zip(metas, chunk)):
                one = {"pred_logits": logits[i : i + 1], "pred_boxes": boxes[i : i + 1]}
                cells, kv_regions, grid_regions = self.postprocess(one, data, table.box)
                if len(cells) == 0:
                    continue
                outputs.append(TableDetectorSchema(id=None, box=table.box, role=table.role, cells=cells, kv_regions=kv_regions,
                                                   grid_regions=grid_regions))
        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        return outputs


__all__ = ["CellDetector", "TableParserModelCatalog", "find_holes_as_rects", "calc_adjacent_holes_to_cells", "choose_role",
           "filter_contained_rectangles_with_category", "filter_contained_rectangles_across_categories", "logger"]

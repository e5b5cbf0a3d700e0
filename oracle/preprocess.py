"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the reference's pre-processing call chains (paths relative to
/root/reference/src/yomitoku), built on oracle/cvlike.py (cv2 semantics, parity unpinned) and on
Pillow itself for the RT-DETR inputs (Pillow 12.2.0 is installed: that part IS the real library).
"""

from __future__ import annotations

import numpy as np
import torch
from PIL import Image

from . import cvlike


def resize_shortest_edge(img, shortest_edge_length, max_length):
    """data/functions.py:196-227."""
    h, w = img.shape[:2]
    scale = shortest_edge_length / min(h, w)
    if h < w:
        new_h, new_w = shortest_edge_length, int(w * scale)
    else:
        new_h, new_w = int(h * scale), shortest_edge_length
    if max(new_h, new_w) > max_length:
        scale = float(max_length) / max(new_h, new_w)
        new_h, new_w = int(new_h * scale), int(new_w * scale)
    neww = max(int(new_w / 32) * 32, 32)
    newh = max(int(new_h / 32) * 32, 32)
    return cvlike.resize_area(img, (neww, newh))


def detector_preprocess(img_bgr, shortest_size=1280, limit_size=1600):
    """TextDetector.preprocess (text_detector.py:99-107) + standardization_image + array_to_tensor
    (data/functions.py:230-264), including the double channel flip and the float64 normalisation."""
    img = img_bgr.copy()[:, :, ::-1].astype(np.float32)
    resized = resize_shortest_edge(img, shortest_size, limit_size)
    x = resized[:, :, ::-1]
    x = x / 255.0
    x = (x - np.array((0.485, 0.456, 0.406))) / np.array((0.229, 0.224, 0.225))
    x = x.astype(np.float32)
    return torch.as_tensor(np.transpose(x, (2, 0, 1)).copy(), dtype=torch.float)[None]


def rtdetr_preprocess(img_bgr, box=None, size=(640, 640)):
    """LayoutParser.preprocess (layout_parser.py:195-199) / TableStructureRecognizer.preprocess
    (table_structure_recognizer.py:169-186): BGR->RGB, crop, T.Resize([h, w]) (PIL bilinear with
    antialiasing), T.ToTensor."""
    rgb = img_bgr[:, :, ::-1]
    if box is not None:
        x1, y1, x2, y2 = map(int, box)
        rgb = rgb[y1:y2, x1:x2, :]
    th, tw = rgb.shape[:2]
    pil = Image.fromarray(np.ascontiguousarray(rgb)).resize((size[1], size[0]), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(pil).copy()).permute(2, 0, 1).to(torch.float32).div(255)
    return t[None], (th, tw)


# ------------------------------------------------------------------ recogniser crops (data/dataset.py:44-129)
def validate_quads(img, quad):
    h, w = img.shape[:2]
    if len(quad) != 4:
        return None
    for p in quad:
        if len(p) != 2:
            return None
    q = np.array(quad, dtype=int)
    if q[:, 0].min() < 0 or q[:, 0].max() > w or q[:, 1].min() < 0 or q[:, 1].max() > h:
        return None
    return True


def extract_roi_with_perspective(img, quad):
    """data/functions.py:301-333."""
    quad = np.array(quad, dtype=np.int64)
    roi = img[int(min(quad[:, 1])) : int(max(quad[:, 1])), int(min(quad[:, 0])) : int(max(quad[:, 0])), :]
    quad[:, 0] -= int(min(quad[:, 0]))
    quad[:, 1] -= int(min(quad[:, 1]))
    width = int(np.linalg.norm(quad[0] - quad[1]))
    height = int(np.linalg.norm(quad[1] - quad[2]))
    M = cvlike.perspective_transform(np.float32(quad), np.float32([[0, 0], [width, 0], [width, height], [0, height]]))
    return cvlike.warp_perspective(roi, M, (width, height))


def calc_resize_without_padding(img, target_size):
    """data/functions.py:353-376."""
    h, w = img.shape[:2]
    scale_w = target_size[1] / w if w > target_size[1] else 1.0
    scale_h = target_size[0] / h if h > target_size[0] else 1.0
    s = min(scale_w, scale_h)
    return max(1, int(h * s)), max(1, int(w * s))


def calc_source_levels(quads, target_height, max_level=3):
    """data/dataset.py:16-41: pyramid level per quad, clip(floor(log2(short side / target height)), 0, max_level)."""
    if len(quads) == 0:
        return np.zeros(0, dtype=int)
    q = np.asarray(quads, dtype=np.float32).reshape(-1, 4, 2)
    short = np.maximum(1.0, np.minimum(np.linalg.norm(q[:, 0] - q[:, 1], axis=1), np.linalg.norm(q[:, 1] - q[:, 2], axis=1)))
    return np.clip(np.floor(np.log2(short / float(target_height))).astype(int), 0, max_level)


def canvas_tensor(roi, img_size, dynamic_width=False, align=8, margin=64):
    """resize_with[_dynamic]_padding (data/functions.py:379-439) + ToTensor + Normalize(0.5, 0.5) of a (rotated) ROI."""
    new_h, new_w = calc_resize_without_padding(roi, img_size)
    resized = cvlike.resize_area(roi, (new_w, new_h)) if (new_h, new_w) != roi.shape[:2] else roi.copy()
    canvas_w = min(img_size[1], ((new_w + margin + align - 1) // align) * align) if dynamic_width else img_size[1]
    canvas = np.zeros((img_size[0], canvas_w, 3), dtype=np.uint8)
    canvas[:new_h, :new_w, :] = resized
    t = torch.from_numpy(canvas).permute(2, 0, 1).to(torch.float32).div(255)
    return (t - 0.5) / 0.5, new_w


def parseq_crop(img_rgb, quad, img_size=(32, 800), dynamic_width=False, align=8, margin=64, with_roi=False):
    """ParseqDataset._preprocess_on (data/dataset.py:105-124) + transform (:55-62):
    -> (tensor 3 x 32 x canvas_w in [-1, 1], content_width[, rotated roi]) or None."""
    if validate_quads(img_rgb, quad) is None:
        return None
    roi = extract_roi_with_perspective(img_rgb, quad)
    h, w = roi.shape[:2]
    if h > 2 * w:
        roi = np.ascontiguousarray(np.rot90(roi, 1))  # cv2.ROTATE_90_COUNTERCLOCKWISE
    new_h, new_w = calc_resize_without_padding(roi, img_size)
    resized = cvlike.resize_area(roi, (new_w, new_h)) if (new_h, new_w) != roi.shape[:2] else roi.copy()
    canvas_w = min(img_size[1], ((new_w + margin + align - 1) // align) * align) if dynamic_width else img_size[1]
    canvas = np.zeros((img_size[0], canvas_w, 3), dtype=np.uint8)
    canvas[:new_h, :new_w, :] = resized
    t = torch.from_numpy(canvas).permute(2, 0, 1).to(torch.float32).div(255)
    t = (t - 0.5) / 0.5
    return (t, new_w, roi) if with_roi else (t, new_w)

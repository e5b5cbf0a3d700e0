"""Integer box geometry used by the post-processing and aggregation stages (the on-path subset of
the reference's utils/misc.py:35-134).  Boxes are [x1, y1, x2, y2]; quads are 4 x [x, y]."""

from __future__ import annotations


def filter_by_flag(elements, flags):
    assert len(elements) == len(flags)
    return [e for e, keep in zip(elements, flags) if keep]


def calc_intersection(rect_a, rect_b):
    """Intersection box of two rectangles (coordinates truncated to int), or None when empty."""
    ax1, ay1, ax2, ay2 = (int(v) for v in rect_a)
    bx1, by1, bx2, by2 = (int(v) for v in rect_b)
    x1, y1 = max(ax1, bx1), max(ay1, by1)
    x2, y2 = min(ax2, bx2), min(ay2, by2)
    if x2 - x1 <= 0 or y2 - y1 <= 0:
        return None
    return [x1, y1, x2, y2]


def calc_overlap_ratio(rect_a, rect_b):
    """(area(A & B) / area(B), intersection) - the share of B covered by A."""
    inter = calc_intersection(rect_a, rect_b)
    if inter is None:
        return 0, None
    bx1, by1, bx2, by2 = rect_b
    ratio = ((inter[2] - inter[0]) * (inter[3] - inter[1])) / ((bx2 - bx1) * (by2 - by1))
    return ratio, inter


def is_contained(rect_a, rect_b, threshold=0.8):
    """True when more than `threshold` of rectangle B lies inside rectangle A."""
    return calc_overlap_ratio(rect_a, rect_b)[0] > threshold


def is_intersected_horizontal(rect_a, rect_b, threshold=0.5):
    """Do the two boxes share at least `threshold` of the shorter one's height (same text row)?"""
    _, ay1, _, ay2 = (int(v) for v in rect_a)
    _, by1, _, by2 = (int(v) for v in rect_b)
    overlap = max(0, min(ay2, by2) - max(ay1, by1))
    return not (overlap / min(ay2 - ay1, by2 - by1)) < threshold


def is_intersected_vertical(rect_a, rect_b):
    """Do the two boxes overlap at all along x (same text column)?"""
    ax1, _, ax2, _ = (int(v) for v in rect_a)
    bx1, _, bx2, _ = (int(v) for v in rect_b)
    return max(0, min(ax2, bx2) - max(ax1, bx1)) != 0


def containment_matrix(boxes_a, boxes_b, threshold):
    """[len(a)][len(b)] bool: is_contained(a, b, threshold) for every pair in one shot.  Same integer
    truncation and the same float64 quotient as the scalar form, so the flags are identical."""
    import numpy as np

    if len(boxes_a) == 0 or len(boxes_b) == 0:
        return np.zeros((len(boxes_a), len(boxes_b)), dtype=bool)
    raw_b = np.asarray(boxes_b, dtype=np.float64).reshape(-1, 4)
    a = np.trunc(np.asarray(boxes_a, dtype=np.float64).reshape(-1, 4)).astype(np.int64)
    b = np.trunc(raw_b).astype(np.int64)
    w = np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0])
    h = np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1])
    hit = (w > 0) & (h > 0)
    area_b = (raw_b[:, 2] - raw_b[:, 0]) * (raw_b[:, 3] - raw_b[:, 1])  # the scalar form divides by the raw extent
    if np.any(hit & (area_b[None, :] == 0)):
        raise ZeroDivisionError("division by zero")  # what the scalar form does for a degenerate box
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = (w * h) / area_b[None, :]
    return hit & (ratio > threshold)


def quad_to_xyxy(quad):
    xs = [p[0] for p in quad]
    ys = [p[1] for p in quad]
    return min(xs), min(ys), max(xs), max(ys)

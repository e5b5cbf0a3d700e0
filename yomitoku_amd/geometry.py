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


# ---------------------------------------------------------------------------------------------- table cell detector
# utils/misc.py:182-441 of the reference: IoU and the "is B the right / bottom neighbour of A" predicates that
# calc_adjacent_holes_to_cells (table_cell_detector.py:161-192) uses; plain float arithmetic, same order.
import math  # noqa: E402


def calc_iou(rect_a, rect_b):
    inter = calc_intersection(rect_a, rect_b)
    if inter is None:
        return 0
    ix1, iy1, ix2, iy2 = inter
    bx1, by1, bx2, by2 = rect_b
    ax1, ay1, ax2, ay2 = rect_a
    overlap = (ix2 - ix1) * (iy2 - iy1)
    return overlap / ((ax2 - ax1) * (ay2 - ay1) + (bx2 - bx1) * (by2 - by1) - overlap)


def point_to_segment_distance(px, py, ax, ay, bx, by):
    abx, aby = bx - ax, by - ay
    apx, apy = px - ax, py - ay
    denom = abx * abx + aby * aby
    if denom == 0:
        return math.hypot(px - ax, py - ay)
    t = max(0.0, min(1.0, (apx * abx + apy * aby) / denom))
    return math.hypot(px - (ax + t * abx), py - (ay + t * aby))


def _edge_distances(p1, p2, q1, q2):
    """(max(d1, d4), max(d2, d3), max(d3, d4), max(d1, d2)) for the facing edges p1-p2 (of A) and q1-q2 (of B)."""
    d1 = point_to_segment_distance(*p1, *q1, *q2)
    d2 = point_to_segment_distance(*p2, *q1, *q2)
    d3 = point_to_segment_distance(*q1, *p1, *p2)
    d4 = point_to_segment_distance(*q2, *p1, *p2)
    return max(d1, d4), max(d2, d3), max(d3, d4), max(d1, d2)


def _overlap_interval(i1, i2, j1, j2):
    return max(0.0, min(i2, j2) - max(i1, j1))


def _adjacent(dists, hard, rule, dist_threshold):
    d1, d2, d3, d4 = dists
    if rule == "hard":
        return hard
    if rule == "soft":
        return d1 < dist_threshold or d2 < dist_threshold or d3 < dist_threshold or d4 < dist_threshold
    if rule == "nest":
        return d3 < dist_threshold
    if rule == "child":
        return (not hard) and d3 < dist_threshold
    return False


def is_right_adjacent(box_a, box_b, dist_threshold=15, overlap_ratio_th=0.1, ignore_dist_threshold=10, rule="soft"):
    """Is box_b the right-hand neighbour of box_a (utils/misc.py:299-352)."""
    ax1, ay1, ax2, ay2 = box_a
    bx1, by1, bx2, by2 = box_b
    if bx1 < ax1:
        return False
    if _overlap_interval(ay1, ay2, by1, by2) < overlap_ratio_th * min(ay2 - ay1, by2 - by1):
        return False
    if math.hypot(ax2 - bx1, ay2 - by1) < ignore_dist_threshold or math.hypot(ax2 - bx1, ay1 - by2) < ignore_dist_threshold:
        return False
    dists = _edge_distances((ax2, ay1), (ax2, ay2), (bx1, by1), (bx1, by2))
    hard = math.hypot(ax2 - bx1, ay1 - by1) < dist_threshold and math.hypot(ax2 - bx1, ay2 - by2) < dist_threshold
    return bool(_adjacent(dists, hard, rule, dist_threshold))


def is_bottom_adjacent(box_a, box_b, dist_threshold=15, overlap_ratio_th=0.1, ignore_dist_threshold=10, rule="soft"):
    """Is box_b the neighbour below box_a (utils/misc.py:355-441)."""
    ax1, ay1, ax2, ay2 = box_a
    bx1, by1, bx2, by2 = box_b
    if by1 < ay1:
        return False
    if _overlap_interval(ax1, ax2, bx1, bx2) < overlap_ratio_th * min(ax2 - ax1, bx2 - bx1):
        return False
    if math.hypot(ax2 - bx1, ay2 - by1) < ignore_dist_threshold or math.hypot(ax1 - bx2, ay2 - by1) < ignore_dist_threshold:
        return False
    dists = _edge_distances((ax1, ay2), (ax2, ay2), (bx1, by1), (bx2, by1))
    hard = math.hypot(ax1 - bx1, ay2 - by1) < dist_threshold and math.hypot(ax2 - bx2, ay2 - by1) < dist_threshold
    return bool(_adjacent(dists, hard, rule, dist_threshold))


# ---- the same two predicates (rule "soft", the one calc_adjacent_holes_to_cells uses) for every pair of two box lists at
# once.  Every distance is the same float64 expression as in the scalar form (numpy's hypot is C hypot, like math.hypot),
# so the flags are identical; tests/test_cells.py compares the two forms on random boxes.
def _p2s_matrix(px, py, ax, ay, bx, by):
    import numpy as np

    abx, aby = bx - ax, by - ay
    apx, apy = px - ax, py - ay
    denom = abx * abx + aby * aby
    with np.errstate(divide="ignore", invalid="ignore"):
        t = np.maximum(0.0, np.minimum(1.0, (apx * abx + apy * aby) / denom))
    t = np.where(denom == 0, 0.0, t)  # a degenerate segment: distance to its one point
    return np.hypot(px - (ax + t * abx), py - (ay + t * aby))


def adjacency_matrices(boxes_a, boxes_b, dist_threshold=15, overlap_ratio_th=0.1, ignore_dist_threshold=10):
    """(right, bottom): right[i][j] = is_right_adjacent(a_i, b_j), bottom[i][j] = is_bottom_adjacent(a_i, b_j), rule "soft"."""
    import numpy as np

    if len(boxes_a) == 0 or len(boxes_b) == 0:
        empty = np.zeros((len(boxes_a), len(boxes_b)), dtype=bool)
        return empty, empty.copy()
    a = np.asarray(boxes_a, dtype=np.float64).reshape(-1, 4)[:, None, :]
    b = np.asarray(boxes_b, dtype=np.float64).reshape(-1, 4)[None, :, :]
    ax1, ay1, ax2, ay2 = (a[..., k] + 0 * b[..., 0] for k in range(4))
    bx1, by1, bx2, by2 = (b[..., k] + 0 * a[..., 0] for k in range(4))

    def soft(p1, p2, q1, q2):
        d1, d2 = _p2s_matrix(*p1, *q1, *q2), _p2s_matrix(*p2, *q1, *q2)
        d3, d4 = _p2s_matrix(*q1, *p1, *p2), _p2s_matrix(*q2, *p1, *p2)
        near = np.maximum(d1, d4) < dist_threshold
        for d in (np.maximum(d2, d3), np.maximum(d3, d4), np.maximum(d1, d2)):
            near |= d < dist_threshold
        return near

    def overlap(i1, i2, j1, j2):
        return np.maximum(0.0, np.minimum(i2, j2) - np.maximum(i1, j1))

    right = ~(bx1 < ax1)
    right &= ~(overlap(ay1, ay2, by1, by2) < overlap_ratio_th * np.minimum(ay2 - ay1, by2 - by1))
    right &= ~((np.hypot(ax2 - bx1, ay2 - by1) < ignore_dist_threshold) | (np.hypot(ax2 - bx1, ay1 - by2) < ignore_dist_threshold))
    right &= soft((ax2, ay1), (ax2, ay2), (bx1, by1), (bx1, by2))
    bottom = ~(by1 < ay1)
    bottom &= ~(overlap(ax1, ax2, bx1, bx2) < overlap_ratio_th * np.minimum(ax2 - ax1, bx2 - bx1))
    bottom &= ~((np.hypot(ax2 - bx1, ay2 - by1) < ignore_dist_threshold) | (np.hypot(ax1 - bx2, ay2 - by1) < ignore_dist_threshold))
    bottom &= soft((ax1, ay2), (ax2, ay2), (bx1, by1), (bx2, by1))
    return right, bottom

"""Reading-order estimation for page elements (reference reading_order.py:14-224).

Builds a "comes before" DAG between boxes that share a column (top2bottom) or a row (right2left /
left2right) with nothing in between, then linearises it with a priority depth-first walk.  The
walk is reproduced decision for decision - including its list-mutation-while-iterating quirks,
which decide the order of siblings - because the north-star demands bit-exact reading order.
Elements are addressed by index; the graph is two adjacency lists.
"""

from __future__ import annotations

import numpy as np

from .geometry import is_intersected_horizontal, is_intersected_vertical


class _Graph:
    def __init__(self, boxes):
        n = len(boxes)
        self.box = boxes
        self.children = [[] for _ in range(n)]
        self.parents = [[] for _ in range(n)]
        self.distance = [0] * n

    def link(self, a, b):
        if b in self.children[a]:
            return
        self.children[a].append(b)
        self.parents[b].append(a)


def _something_between(g, a, b, axis):
    """Is a third box strictly between a and b along `axis` while overlapping a on the other axis?
    axis 1: vertical gap test (top2bottom), axis 0: horizontal gap test."""
    lo, hi = axis, axis + 2
    a_lo, a_hi = g.box[a][lo], g.box[a][hi]
    b_lo, b_hi = g.box[b][lo], g.box[b][hi]
    shares = is_intersected_vertical if axis == 1 else is_intersected_horizontal
    for s in range(len(g.box)):
        if s == a or s == b:
            continue
        if not shares(g.box[s], g.box[a]):
            continue
        s_lo, s_hi = g.box[s][lo], g.box[s][hi]
        if a_hi < s_lo < b_lo and a_hi < s_hi < b_lo:
            return True
        if b_hi < s_lo < a_lo and b_hi < s_hi < a_lo:
            return True
    return False


VECTOR_MIN_BOXES = 6  # below this the scalar double loop is cheaper than the numpy set-up


def _adjacent_pairs(boxes, axis):
    """All (i, j), i != j, in row-major order, for which the scalar loop's test
    `shares(box i, box j) and not _something_between(i, j)` holds - computed with numpy: an n x n
    share matrix, then the in-between test only for the sharing pairs (P x n).  Returns None when a
    degenerate box would make the scalar form raise, so that the caller takes that path and raises too."""
    raw = np.asarray(boxes, dtype=np.float64)
    t = np.trunc(raw).astype(np.int64)
    n = len(boxes)
    if axis == 1:  # is_intersected_vertical: any overlap along x
        ov = np.minimum(t[:, None, 2], t[None, :, 2]) - np.maximum(t[:, None, 0], t[None, :, 0])
        share = np.maximum(ov, 0) != 0
    else:  # is_intersected_horizontal: >= half of the shorter box's height
        h = t[:, 3] - t[:, 1]
        if np.any(h == 0):
            return None
        ov = np.maximum(0, np.minimum(t[:, None, 3], t[None, :, 3]) - np.maximum(t[:, None, 1], t[None, :, 1]))
        share = ~((ov / np.minimum(h[:, None], h[None, :])) < 0.5)
    np.fill_diagonal(share, False)
    pa, pb = np.nonzero(share)  # row-major: the order the scalar loops visit the pairs in
    if len(pa) == 0:
        return []
    lo, hi = raw[:, axis], raw[:, axis + 2]
    a_lo, a_hi, b_lo, b_hi = lo[pa, None], hi[pa, None], lo[pb, None], hi[pb, None]
    s_lo, s_hi = lo[None, :], hi[None, :]
    gap = ((a_hi < s_lo) & (s_lo < b_lo) & (a_hi < s_hi) & (s_hi < b_lo)) | \
          ((b_hi < s_lo) & (s_lo < a_lo) & (b_hi < s_hi) & (s_hi < a_lo))
    gap &= share[pa]  # share is symmetric: third box s overlaps a; its diagonal already excludes s == a
    gap[np.arange(len(pb)), pb] = False  # s == b
    keep = ~gap.any(axis=1)
    return list(zip(pa[keep].tolist(), pb[keep].tolist()))


def _build(boxes, direction):
    g = _Graph(boxes)
    n = len(boxes)
    pairs = None
    if n >= VECTOR_MIN_BOXES and direction in ("top2bottom", "right2left", "left2right"):
        pairs = _adjacent_pairs(boxes, 1 if direction == "top2bottom" else 0)
    if pairs is not None:
        if direction == "top2bottom":
            for i, j in pairs:
                if boxes[i][1] < boxes[j][1]:
                    g.link(i, j)
                else:
                    g.link(j, i)
            for i in range(n):
                g.distance[i] = boxes[i][0] + boxes[i][1]
            sib_key = 0
        else:
            max_x = max(b[2] for b in boxes)
            for i, j in pairs:
                ti, tj = boxes[i][2], boxes[j][2]
                if direction == "right2left":
                    first, second = (j, i) if ti < tj else (i, j)
                else:
                    first, second = (j, i) if tj < ti else (i, j)
                g.link(first, second)
            for i in range(n):
                if direction == "right2left":
                    g.distance[i] = (max_x - boxes[i][2]) + boxes[i][1]
                else:
                    g.distance[i] = boxes[i][0] * 1 + boxes[i][1] * 5
            sib_key = 1
    elif direction == "top2bottom":
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                if is_intersected_vertical(boxes[i], boxes[j]) and not _something_between(g, i, j, 1):
                    if boxes[i][1] < boxes[j][1]:
                        g.link(i, j)
                    else:
                        g.link(j, i)
            g.distance[i] = boxes[i][0] + boxes[i][1]
        sib_key = 0
    elif direction in ("right2left", "left2right"):
        max_x = max(b[2] for b in boxes)
        for i in range(n):
            for j in range(n):
                if i == j:
                    continue
                if is_intersected_horizontal(boxes[i], boxes[j]) and not _something_between(g, i, j, 0):
                    ti, tj = boxes[i][2], boxes[j][2]
                    if direction == "right2left":
                        first, second = (j, i) if ti < tj else (i, j)
                    else:
                        first, second = (j, i) if tj < ti else (i, j)
                    g.link(first, second)
            if direction == "right2left":
                g.distance[i] = (max_x - boxes[i][2]) + boxes[i][1]
            else:
                g.distance[i] = boxes[i][0] * 1 + boxes[i][1] * 5
        sib_key = 1
    else:
        raise ValueError(f"Invalid direction: {direction}")
    for i in range(n):
        g.children[i] = sorted(g.children[i], key=lambda c: boxes[c][sib_key])
    return g


def _walk(g, direction):
    n = len(g.box)
    if n == 0:
        return []
    pending = sorted(range(n), key=lambda i: g.distance[i])
    visited = [False] * n
    stack = [pending.pop(0)]
    order = []
    parked = []  # nodes met before all their parents were emitted
    sib_axis = 0 if direction == "top2bottom" else 1
    while not all(visited):
        while stack:
            cur = stack.pop()
            emitted = False
            if not visited[cur]:
                if all(visited[p] for p in g.parents[cur]):
                    visited[cur] = True
                    order.append(cur)
                    emitted = True
                elif cur not in parked:
                    parked.append(cur)
            if emitted:
                while parked:  # re-queue parked nodes, last parked first
                    stack.append(parked.pop())
            if g.children[cur]:
                stack.append(cur)
                stack.append(g.children[cur].pop(0))
                continue
            # leaf (or exhausted node): pull its not-yet-walked children forward.  The reference removes
            # from the list it is iterating, so the element following each removal is skipped; keep that.
            pulled = []
            it = 0
            while it < len(stack):
                cand = stack[it]
                if cur in g.parents[cand]:
                    pulled.append(cand)
                    stack.remove(cand)  # first occurrence, like list.remove
                it += 1
            pulled.sort(key=lambda c: g.box[c][sib_axis], reverse=True)
            stack.extend(pulled)
        for cand in pending:
            if cand in parked:
                continue
            stack.append(cand)
            pending.remove(cand)
            break
        else:
            if not all(visited) and parked:
                forced = parked.pop(0)
                visited[forced] = True
                order.append(forced)
    return order


def prediction_reading_order(elements, direction, img=None):
    """Assign `.order` (0-based rank) to every element in place and return the list."""
    if len(elements) < 2:
        return elements
    boxes = [list(e.box) for e in elements]
    g = _build(boxes, direction)
    for rank, idx in enumerate(_walk(g, direction)):
        elements[idx].order = rank
    return elements

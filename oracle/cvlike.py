"""ORACLE (test infrastructure only - never imported by the product path).

NumPy / pure-Python restatement of the third-party image routines the reference calls on the hot
path.  None of these libraries is installed in the build image (opencv-python 4.13.0.92
uv.lock:1518, pyclipper 1.4.0 uv.lock:1678, shapely 2.1.2 uv.lock:2241), so this file follows their
published algorithms and the call sites / arguments in the reference:
**parity unpinned** against the libraries themselves; the reference's own tests pin only shapes
(tests/test_data.py:83-138), which tests/test_imaging.py re-checks.

  resize_area            cv2.resize(..., INTER_AREA)         data/functions.py:226,393,431
  find_borders           cv2.findContours(RETR_LIST, ...)    dbnet_postporcessor.py:43-47
  min_area_rect          cv2.minAreaRect + cv2.boxPoints     dbnet_postporcessor.py:101-102
  polygon_mean           cv2.fillPoly + cv2.mean             dbnet_postporcessor.py:126-138
  offset_round           pyclipper JT_ROUND closed offset    dbnet_postporcessor.py:95-98
  perspective_transform / warp_perspective                   data/functions.py:331-332
"""

from __future__ import annotations

import math

import numpy as np

# ------------------------------------------------------------------------------------------
# cv2.resize INTER_AREA


def _area_tab(ssize, dsize, scale):
    """computeResizeAreaTab: (dst index, src index, weight) triples with float32 weights."""
    tab = []
    for dx in range(dsize):
        fsx1 = dx * scale
        fsx2 = fsx1 + scale
        cell = min(scale, ssize - fsx1)
        sx1, sx2 = math.ceil(fsx1), math.floor(fsx2)
        sx2 = min(sx2, ssize - 1)
        sx1 = min(sx1, sx2)
        if sx1 - fsx1 > 1e-3:
            tab.append((dx, sx1 - 1, np.float32((sx1 - fsx1) / cell)))
        for sx in range(sx1, sx2):
            tab.append((dx, sx, np.float32(1.0 / cell)))
        if fsx2 - sx2 > 1e-3:
            tab.append((dx, sx2, np.float32(min(min(fsx2 - sx2, 1.0), cell) / cell)))
    return tab


def _linear_area_coeffs(ssize, dsize, scale):
    """INTER_AREA when enlarging: bilinear taps with the 'area' coefficient rule (resize.cpp)."""
    inv = 1.0 / scale
    idx = np.zeros(dsize, dtype=np.int64)
    frac = np.zeros(dsize, dtype=np.float32)
    for dx in range(dsize):
        sx = math.floor(dx * scale)
        fx = np.float32((dx + 1) - (sx + 1) * inv)
        fx = np.float32(0.0) if fx <= 0 else np.float32(fx - math.floor(fx))
        if sx < 0:
            fx, sx = np.float32(0.0), 0
        if sx >= ssize - 1:
            fx, sx = np.float32(0.0), ssize - 1
        idx[dx], frac[dx] = sx, fx
    return idx, frac


def resize_area(img: np.ndarray, dsize_wh) -> np.ndarray:
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_AREA) for float32 or uint8 HxWxC."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = img.shape[:2]
    is_u8 = img.dtype == np.uint8
    src = img.astype(np.float32)
    if src.ndim == 2:
        src = src[:, :, None]
    scale_x, scale_y = sw / dw, sh / dh
    if scale_x >= 1 and scale_y >= 1:
        ix, iy = round(scale_x), round(scale_y)
        if is_u8 and abs(scale_x - ix) < 2.3e-16 and abs(scale_y - iy) < 2.3e-16 and not (ix == 1 and iy == 1):
            # ResizeAreaFast: integer block sums; 2x2 rounds half up, other factors rint(sum * (1.f / area))
            s = img.astype(np.int32).reshape(dh, iy, dw, ix, -1).sum(axis=(1, 3))
            if ix == 2 and iy == 2:
                out = (s + 2) >> 2
            else:
                out = np.clip(np.rint(s.astype(np.float32) * np.float32(1.0 / (ix * iy))), 0, 255)
            return out.astype(np.uint8).reshape((dh, dw) + img.shape[2:])
        xtab, ytab = _area_tab(sw, dw, scale_x), _area_tab(sh, dh, scale_y)
        # horizontal pass per source row (float32 accumulation in table order)
        buf = np.zeros((sh, dw, src.shape[2]), dtype=np.float32)
        for dx, sx, a in xtab:
            buf[:, dx] = buf[:, dx] + src[:, sx] * a
        out = np.zeros((dh, dw, src.shape[2]), dtype=np.float32)
        first = [True] * dh
        for dy, sy, b in ytab:
            if first[dy]:
                out[dy] = b * buf[sy]
                first[dy] = False
            else:
                out[dy] = out[dy] + b * buf[sy]
    else:
        xi, xf = _linear_area_coeffs(sw, dw, scale_x)
        yi, yf = _linear_area_coeffs(sh, dh, scale_y)
        x1 = np.minimum(xi + 1, sw - 1)
        y1 = np.minimum(yi + 1, sh - 1)
        a0, a1 = (np.float32(1.0) - xf)[None, :, None], xf[None, :, None]
        rows = src[:, xi] * a0 + src[:, x1] * a1  # HResizeLinear
        b0, b1 = (np.float32(1.0) - yf)[:, None, None], yf[:, None, None]
        out = rows[yi] * b0 + rows[y1] * b1  # VResizeLinear
    if is_u8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)  # saturate_cast<uchar>: round half to even
    return out.reshape((dh, dw) + img.shape[2:])


def resize_half(img: np.ndarray) -> np.ndarray:
    """cv2.resize(img, None, fx=0.5, fy=0.5, interpolation=cv2.INTER_AREA) on uint8 HxWx3 (data/dataset.py:76-78).
    dsize = saturate_cast<int>(size * 0.5) (round half to even); the scale stays exactly 2, so this is
    ResizeAreaFast: full 2x2 cells round as (s + 2) >> 2, cells cut by the right / bottom edge average the pixels
    they have (saturate_cast<uchar>((float)sum / count)), cells starting past the edge are 0.
    OpenCV 4.x imgproc/resize.cpp, restated from the published source; parity unpinned (cv2 is not installed)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    sh, sw = img.shape[:2]
    dh, dw = int(round(sh * 0.5)), int(round(sw * 0.5))
    out = np.zeros((dh, dw, img.shape[2]), dtype=np.uint8)
    fh, fw = min(dh, sh // 2), min(dw, sw // 2)  # rows / columns of complete cells
    s = img[: 2 * fh, : 2 * fw].astype(np.int32).reshape(fh, 2, fw, 2, -1).sum(axis=(1, 3))
    out[:fh, :fw] = ((s + 2) >> 2).astype(np.uint8)
    for dy in range(dh):
        for dx in range(dw):
            if dy < fh and dx < fw:
                continue
            sy0, sx0 = 2 * dy, 2 * dx
            if sy0 >= sh or sx0 >= sw:
                continue
            cell = img[sy0 : min(sy0 + 2, sh), sx0 : min(sx0 + 2, sw)].astype(np.int32).reshape(-1, img.shape[2])
            val = np.rint(cell.sum(0).astype(np.float32) / np.float32(cell.shape[0]))
            out[dy, dx] = np.clip(val, 0, 255).astype(np.uint8)
    return out


# ------------------------------------------------------------------------------------------
# cv2.findContours(RETR_LIST): Suzuki & Abe border following, 8-connected foreground
_NB = [(0, 1), (1, 1), (1, 0), (1, -1), (0, -1), (-1, -1), (-1, 0), (-1, 1)]  # clockwise from east (dy, dx)
_NB_INDEX = {d: i for i, d in enumerate(_NB)}


def find_borders(bitmap: np.ndarray):
    """All outer and hole borders of a binary image as closed pixel chains [(x, y), ...], newest
    first (the order cv2 hands RETR_LIST contours back)."""
    h, w = bitmap.shape
    f = np.zeros((h + 2, w + 2), dtype=np.int32)
    f[1:-1, 1:-1] = (bitmap != 0).astype(np.int32)
    nbd = 1
    found = []
    for i in range(1, h + 1):
        row = f[i]
        js = np.nonzero(row)[0]
        for j in js:
            v = row[j]
            if v == 1 and row[j - 1] == 0:
                start = (0, -1)
            elif v >= 1 and row[j + 1] == 0:
                start = (0, 1)
            else:
                continue
            nbd += 1
            d0 = _NB_INDEX[start]
            d1 = None
            for k in range(8):
                d = (d0 + k) % 8
                if f[i + _NB[d][0], j + _NB[d][1]] != 0:
                    d1 = d
                    break
            if d1 is None:
                f[i, j] = -nbd
                found.append([(j - 1, i - 1)])
                continue
            i1, j1 = i + _NB[d1][0], j + _NB[d1][1]
            pi, pj, ci, cj = i1, j1, i, j
            chain = []
            while True:
                ds = _NB_INDEX[(pi - ci, pj - cj)]
                east_zero = False
                for k in range(1, 9):
                    d = (ds - k) % 8
                    if f[ci + _NB[d][0], cj + _NB[d][1]] != 0:
                        dn = d
                        break
                    if d == 0:
                        east_zero = True
                chain.append((cj - 1, ci - 1))
                if east_zero:
                    f[ci, cj] = -nbd
                elif f[ci, cj] == 1:
                    f[ci, cj] = nbd
                ni, nj = ci + _NB[dn][0], cj + _NB[dn][1]
                if (ni, nj) == (i, j) and (ci, cj) == (i1, j1):
                    break
                pi, pj, ci, cj = ci, cj, ni, nj
            found.append(chain)
    return found[::-1]


# ------------------------------------------------------------------------------------------
# cv2.minAreaRect + cv2.boxPoints
def _hull(points):
    pts = sorted(set((int(x), int(y)) for x, y in points))
    if len(pts) < 3:
        return pts

    def cross(o, a, b):
        return (a[0] - o[0]) * (b[1] - o[1]) - (a[1] - o[1]) * (b[0] - o[0])

    lower, upper = [], []
    for p in pts:
        while len(lower) >= 2 and cross(lower[-2], lower[-1], p) <= 0:
            lower.pop()
        lower.append(p)
    for p in reversed(pts):
        while len(upper) >= 2 and cross(upper[-2], upper[-1], p) <= 0:
            upper.pop()
        upper.append(p)
    return lower[:-1] + upper[:-1]


def min_area_rect(points):
    """(4 corner points float32 [4,2], short side) of the minimum-area enclosing rectangle."""
    hull = _hull(points)
    n = len(hull)
    if n == 1:
        return np.array([hull[0]] * 4, dtype=np.float32), 0.0
    if n == 2:
        return np.array([hull[0], hull[1], hull[1], hull[0]], dtype=np.float32), 0.0
    hp = np.array(hull, dtype=np.float64)
    best = None
    for e in range(n):
        a, b = hp[e], hp[(e + 1) % n]
        d = b - a
        ln = math.hypot(d[0], d[1])
        if ln == 0:
            continue
        ux, uy = d[0] / ln, d[1] / ln
        u = hp[:, 0] * ux + hp[:, 1] * uy
        v = -hp[:, 0] * uy + hp[:, 1] * ux
        area = (u.max() - u.min()) * (v.max() - v.min())
        if best is None or area < best[0]:
            us = [u.min(), u.max(), u.max(), u.min()]
            vs = [v.min(), v.min(), v.max(), v.max()]
            corners = [(us[c] * ux - vs[c] * uy, us[c] * uy + vs[c] * ux) for c in range(4)]
            best = (area, corners, min(u.max() - u.min(), v.max() - v.min()))
    return np.array(best[1], dtype=np.float32), float(np.float32(best[2]))


# ------------------------------------------------------------------------------------------
# cv2.fillPoly(mask, contour) + cv2.mean(pred, mask)
def polygon_mean(pred: np.ndarray, chain) -> float:
    h, w = pred.shape
    pts = np.array(chain, dtype=np.int64)
    xmin, xmax = int(np.clip(pts[:, 0].min(), 0, w - 1)), int(np.clip(pts[:, 0].max(), 0, w - 1))
    ymin, ymax = int(np.clip(pts[:, 1].min(), 0, h - 1)), int(np.clip(pts[:, 1].max(), 0, h - 1))
    mask = np.zeros((ymax - ymin + 1, xmax - xmin + 1), dtype=bool)
    mask[pts[:, 1] - ymin, pts[:, 0] - xmin] = True
    n = len(chain)
    for y in range(ymin, ymax + 1):
        xs = []
        for e in range(n):
            (ax, ay), (bx, by) = chain[e], chain[(e + 1) % n]
            if (ay <= y) == (by <= y):
                continue
            xs.append(ax + (y - ay) * (bx - ax) / (by - ay))
        xs.sort()
        for k in range(0, len(xs) - 1, 2):
            x0, x1 = max(math.floor(xs[k]) + 1, xmin), min(math.ceil(xs[k + 1]) - 1, xmax)
            if x1 >= x0:
                mask[y - ymin, x0 - xmin : x1 - xmin + 1] = True
    roi = pred[ymin : ymax + 1, xmin : xmax + 1]
    return float(roi[mask].astype(np.float64).sum() / mask.sum())


# ------------------------------------------------------------------------------------------
# pyclipper.PyclipperOffset().AddPath(box, JT_ROUND, ET_CLOSEDPOLYGON); Execute(distance)  (Clipper 6.4.2)
def _cround(v):
    return int(v - 0.5) if v < 0 else int(v + 0.5)


def offset_round(path, delta, arc_tolerance=0.25):
    src = []
    pts = [(int(p[0]), int(p[1])) for p in path]  # the binding truncates to integers
    hi = len(pts) - 1
    while hi > 0 and pts[0] == pts[hi]:
        hi -= 1
    for p in pts[: hi + 1]:
        if not src or src[-1] != p:
            src.append(p)
    n = len(src)
    if n < 3:
        return []
    a = 0.0
    j = n - 1
    for i in range(n):
        a += (src[j][0] + src[i][0]) * (src[j][1] - src[i][1])
        j = i
    if -a * 0.5 < 0:
        src.reverse()
    # ClipperOffset::DoOffset: the arc tolerance is capped at |delta| * 0.25 (matters for |delta| < 1 only)
    y = min(arc_tolerance, abs(delta) * 0.25)
    steps = math.pi / math.acos(1 - y / abs(delta))
    steps = min(steps, abs(delta) * math.pi)
    m_sin, m_cos = math.sin(2 * math.pi / steps), math.cos(2 * math.pi / steps)
    per_rad = steps / (2 * math.pi)
    if delta < 0:
        m_sin = -m_sin
    normals = []
    for j in range(n):
        dx, dy = src[(j + 1) % n][0] - src[j][0], src[(j + 1) % n][1] - src[j][1]
        f = 1.0 / math.sqrt(dx * dx + dy * dy)
        normals.append((dy * f, -dx * f))
    out = []
    k = n - 1
    for j in range(n):
        sin_a = normals[k][0] * normals[j][1] - normals[j][0] * normals[k][1]
        done = False
        if abs(sin_a * delta) < 1.0:
            if normals[k][0] * normals[j][0] + normals[j][1] * normals[k][1] > 0:
                out.append((_cround(src[j][0] + normals[k][0] * delta), _cround(src[j][1] + normals[k][1] * delta)))
                done = True
        else:
            sin_a = max(-1.0, min(1.0, sin_a))
        if not done:
            if sin_a * delta < 0:
                out.append((_cround(src[j][0] + normals[k][0] * delta), _cround(src[j][1] + normals[k][1] * delta)))
                out.append(src[j])
                out.append((_cround(src[j][0] + normals[j][0] * delta), _cround(src[j][1] + normals[j][1] * delta)))
            else:
                ang = math.atan2(sin_a, normals[k][0] * normals[j][0] + normals[k][1] * normals[j][1])
                st = max(_cround(per_rad * abs(ang)), 1)
                X, Y = normals[k]
                for _ in range(st):
                    out.append((_cround(src[j][0] + X * delta), _cround(src[j][1] + Y * delta)))
                    X, Y = X * m_cos - m_sin * Y, X * m_sin + Y * m_cos
                out.append((_cround(src[j][0] + normals[j][0] * delta), _cround(src[j][1] + normals[j][1] * delta)))
        k = j
    return out


# ------------------------------------------------------------------------------------------
# the post-processor itself (postprocessor/dbnet_postporcessor.py:8-138)
def _order_box(pts):
    p = sorted(list(pts), key=lambda q: q[0])
    i1, i4 = (0, 1) if p[1][1] > p[0][1] else (1, 0)
    i2, i3 = (2, 3) if p[3][1] > p[2][1] else (3, 2)
    return [p[i1], p[i2], p[i3], p[i4]]


def db_postprocess(pred: np.ndarray, image_size, min_size, thresh, box_thresh, max_candidates, unclip_ratio):
    """pred: HxW float32 probability map; image_size = (orig_h, orig_w) -> (quads, scores)."""
    height, width = pred.shape
    dest_height, dest_width = image_size
    borders = find_borders(pred > thresh)
    boxes, scores = [], []
    for chain in borders[: min(len(borders), max_candidates)]:
        pts, sside = min_area_rect(chain)
        if sside < min_size:
            continue
        points = np.array(_order_box(pts))
        score = polygon_mean(pred, chain)
        if box_thresh > score:
            continue
        w_ = points[:, 0].max() - points[:, 0].min()
        h_ = points[:, 1].max() - points[:, 1].min()
        area = 0.0
        length = 0.0
        for i in range(4):
            a, b = points[i].astype(np.float64), points[(i + 1) % 4].astype(np.float64)
            area += a[0] * b[1] - b[0] * a[1]
            length += math.hypot(b[0] - a[0], b[1] - a[1])
        area = abs(area) * 0.5
        distance = area * (unclip_ratio / math.sqrt(min(w_, h_))) / length
        grown = offset_round(points, distance)
        if not grown:
            continue
        pts2, sside = min_area_rect(grown)
        if sside < min_size + 2:
            continue
        box = np.array(_order_box(pts2))
        box[:, 0] = np.clip(np.round(box[:, 0] / width * dest_width), 0, dest_width)
        box[:, 1] = np.clip(np.round(box[:, 1] / height * dest_height), 0, dest_height)
        boxes.append(box.astype(np.int16).tolist())
        scores.append(score)
    return boxes, scores


# ------------------------------------------------------------------------------------------
# cv2.getPerspectiveTransform / cv2.warpPerspective(INTER_LINEAR, BORDER_CONSTANT 0) on uint8
def perspective_transform(src4, dst4):
    """3x3 matrix M (float64) with dst ~ M src, solved like cv2.getPerspectiveTransform (LU on the 8x8 system)."""
    a = np.zeros((8, 8), dtype=np.float64)
    b = np.zeros(8, dtype=np.float64)
    for i in range(4):
        x, y = float(src4[i][0]), float(src4[i][1])
        u, v = float(dst4[i][0]), float(dst4[i][1])
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    m = np.linalg.solve(a, b)
    return np.append(m, 1.0).reshape(3, 3)


def warp_perspective(img: np.ndarray, M: np.ndarray, dsize_wh) -> np.ndarray:
    """cv2.warpPerspective(img, M, (w, h)) for uint8 HxWxC: inverse map, source coordinates
    quantised to 1/32 px, bilinear weights in 15-bit fixed point (INTER_BITS=5, INTER_REMAP_COEF_BITS=15),
    constant zero border."""
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    sh, sw = img.shape[:2]
    out = np.zeros((dh, dw, img.shape[2]), dtype=np.uint8)
    if dw == 0 or dh == 0:
        return out
    Mi = np.linalg.inv(M)
    INTER_BITS, TAB = 5, 32
    xs = np.arange(dw, dtype=np.float64)
    for y in range(dh):
        X0 = Mi[0, 0] * xs + Mi[0, 1] * y + Mi[0, 2]
        Y0 = Mi[1, 0] * xs + Mi[1, 1] * y + Mi[1, 2]
        W = Mi[2, 0] * xs + Mi[2, 1] * y + Mi[2, 2]
        W = np.where(W != 0, TAB / np.where(W != 0, W, 1.0), 0.0)
        fX = np.clip(X0 * W, -2147483648.0, 2147483647.0)
        fY = np.clip(Y0 * W, -2147483648.0, 2147483647.0)
        X = np.rint(fX).astype(np.int64)
        Y = np.rint(fY).astype(np.int64)
        sx, sy = X >> INTER_BITS, Y >> INTER_BITS
        ax, ay = (X & (TAB - 1)).astype(np.float32) / TAB, (Y & (TAB - 1)).astype(np.float32) / TAB
        w00 = np.rint((1 - ax) * (1 - ay) * 32768).astype(np.int64)
        w01 = np.rint(ax * (1 - ay) * 32768).astype(np.int64)
        w10 = np.rint((1 - ax) * ay * 32768).astype(np.int64)
        w11 = 32768 - w00 - w01 - w10  # remap's table rows are normalised to sum to 1 << 15

        def tap(yy, xx):
            ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
            v = img[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)].astype(np.int64)
            return np.where(ok[:, None], v, 0)

        acc = (tap(sy, sx) * w00[:, None] + tap(sy, sx + 1) * w01[:, None] + tap(sy + 1, sx) * w10[:, None]
               + tap(sy + 1, sx + 1) * w11[:, None])
        out[y] = np.clip((acc + (1 << 14)) >> 15, 0, 255).astype(np.uint8)
    return out

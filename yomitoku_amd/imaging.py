"""Host-side geometry for the pre-processing kernels: everything here is integer / small-matrix
work on the CPU (sizes, coefficient tables, per-quad warp matrices); the pixels never leave HBM.

Mirrors data/functions.py:196-439 and data/dataset.py:105-124 of the reference for the geometry,
and Pillow's Resample.c for the bilinear coefficient tables.
"""

from __future__ import annotations

import ctypes
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from . import _lib


# ---------------------------------------------------------------------------------------------
def resize_shortest_edge_dims(h: int, w: int, shortest_edge_length: int, max_length: int):
    """Output (height, width) of resize_shortest_edge (data/functions.py:196-227)."""
    scale = shortest_edge_length / min(h, w)
    if h < w:
        new_h, new_w = shortest_edge_length, int(w * scale)
    else:
        new_h, new_w = int(h * scale), shortest_edge_length
    if max(new_h, new_w) > max_length:
        scale = float(max_length) / max(new_h, new_w)
        new_h, new_w = int(new_h * scale), int(new_w * scale)
    return max(int(new_h / 32) * 32, 32), max(int(new_w / 32) * 32, 32)


def page_to_device(img: np.ndarray, device) -> torch.Tensor:
    """uint8 H x W x 3 BGR page -> contiguous device tensor: one H2D copy per page through the device's pinned staging
    ring on the copy stream (yomitoku_amd.data.PageStager); the caller's current stream waits on the copy's event, the
    host does not."""
    from .data.staging import default_stager

    return default_stager(device).upload(img)


def detector_tensor(page_dev: torch.Tensor, shortest: int, limit: int, out: torch.Tensor = None) -> torch.Tensor:
    """TextDetector.preprocess on the device: fp32 1 x 3 x H' x W' (or into `out`, a 3 x H' x W' slice of a batch)."""
    h, w = page_dev.shape[:2]
    oh, ow = resize_shortest_edge_dims(h, w, shortest, limit)
    if out is None:
        out = torch.empty((1, 3, oh, ow), dtype=torch.float32, device=page_dev.device)
    elif tuple(out.shape[-3:]) != (3, oh, ow) or not out.is_contiguous():
        raise ValueError(f"detector_tensor: out must be a contiguous 3 x {oh} x {ow} tensor")
    lib = _lib.load()
    with torch.cuda.device(page_dev.device):
        _lib.check(
            lib.ymk_det_preprocess(page_dev.data_ptr(), h, w, oh, ow, out.data_ptr(), _lib.current_stream_ptr()),
            "ymk_det_preprocess",
        )
    return out


# ---------------------------------------------------------------------------------------------
# Pillow bilinear resample coefficients (Resample.c: precompute_coeffs + normalize_coeffs_8bpc)
def pil_bilinear_coeffs(in_size: int, out_size: int):
    """(bounds int32 [out][2], coefs int32 [out][ksize], ksize).  All outputs at once in float64: the same operations in the
    same order as the per-output loop of Resample.c (`_pil_bilinear_coeffs_scalar`, which tests hold this form to) - the
    weights of an output are summed tap by tap, never pairwise.  A table crop has its own (width, height), so a wave of
    table-heavy pages asks for ~340 new tables: as Python loops that was ~3 ms per crop and THE cost of the tables stage."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    center = (np.arange(out_size, dtype=np.float64) + 0.5) * scale
    xmin = np.maximum((center - support + 0.5).astype(np.int64), 0)          # (int) truncates; the operand is > -1
    xmax = np.minimum((center + support + 0.5).astype(np.int64), in_size) - xmin
    taps = np.arange(ksize, dtype=np.int64)[None, :]
    inside = taps < xmax[:, None]
    a = np.abs((taps + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss
    w = np.where(inside & (a < 1.0), 1.0 - a, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):  # sequential, like `ww += w` in the C loop (taps beyond xmax add an exact 0.0)
        ww = ww + w[:, x]
    k = np.where((ww != 0.0)[:, None], w / np.where(ww != 0.0, ww, 1.0)[:, None], w)
    coefs = np.where(k < 0, (-0.5 + k * float(1 << 22)), (0.5 + k * float(1 << 22))).astype(np.int64).astype(np.int32)
    coefs[~inside] = 0
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, coefs, ksize


def _pil_bilinear_coeffs_scalar(in_size: int, out_size: int):
    """One output at a time, statement for statement after precompute_coeffs / normalize_coeffs_8bpc (the array form above is
    tested against it, tests/test_imaging_host.py)."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    coefs = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        k = []
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - a if a < 1.0 else 0.0
            k.append(w)
            ww += w
        if ww != 0.0:
            k = [v / ww for v in k]
        for x, v in enumerate(k):
            coefs[xx, x] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[xx] = (xmin, xmax)
    return bounds, coefs, ksize


# Coefficient tables on the device, per (sizes, device, STREAM): a table is allocated and uploaded on the stream that
# first needs it and only ever used by kernels queued on that same stream, so the caching allocator's stream-ordered
# reuse rules hold without record_stream; each stream's cache is a small LRU (table crops bring a new (width, 640)
# pair per table).
_COEF_CACHE = {}
_COEF_CACHE_LIMIT = 512
_COEF_LOCK = __import__("threading").Lock()


def _coeffs_on_device(in_size, out_size, device):
    from collections import OrderedDict

    stream = torch.cuda.current_stream(device).cuda_stream if torch.device(device).type == "cuda" else 0
    with _COEF_LOCK:
        cache = _COEF_CACHE.setdefault((str(device), stream), OrderedDict())
        hit = cache.get((in_size, out_size))
        if hit is not None:
            cache.move_to_end((in_size, out_size))
            return hit
    b, c, k = pil_bilinear_coeffs(in_size, out_size)
    hit = (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), k)
    with _COEF_LOCK:
        cache[(in_size, out_size)] = hit
        while len(cache) > _COEF_CACHE_LIMIT:
            cache.popitem(last=False)  # least recently used; its last kernel was queued on this stream before any reuse
    return hit


def rtdetr_tensor(page_dev: torch.Tensor, box: Optional[Sequence[int]], out_hw=(640, 640), out: torch.Tensor = None):
    """LayoutParser / TableStructureRecognizer preprocess on the device: crop `box` (x1, y1, x2, y2) or
    the whole page, BGR->RGB, PIL-style antialiased bilinear resize, /255 -> fp32 3 x H x W."""
    H, W = page_dev.shape[:2]
    if box is None:
        x1, y1, x2, y2 = 0, 0, W, H
    else:
        x1, y1, x2, y2 = (int(v) for v in box)
        # numpy slicing semantics of img[y1:y2, x1:x2] for in-range non-negative indices
        x1, y1, x2, y2 = max(x1, 0), max(y1, 0), min(x2, W), min(y2, H)
    cw, ch = x2 - x1, y2 - y1
    if cw <= 0 or ch <= 0:
        raise ValueError(f"empty crop {box}")
    oh, ow = out_hw
    xb, xk, ksx = _coeffs_on_device(cw, ow, page_dev.device)
    yb, yk, ksy = _coeffs_on_device(ch, oh, page_dev.device)
    if out is None:
        out = torch.empty((3, oh, ow), dtype=torch.float32, device=page_dev.device)
    lib = _lib.load()
    with torch.cuda.device(page_dev.device):
        _lib.check(
            lib.ymk_pil_resize_to_chw(page_dev.data_ptr(), W, x1, y1, xb.data_ptr(), xk.data_ptr(), ksx, yb.data_ptr(),
                                      yk.data_ptr(), ksy, oh, ow, out.data_ptr(), _lib.current_stream_ptr()),
            "ymk_pil_resize_to_chw",
        )
    return out, (ch, cw), (x1, y1)


def _clamped_box(page_dev, box):
    H, W = page_dev.shape[:2]
    if box is None:
        return 0, 0, W, H
    x1, y1, x2, y2 = (int(v) for v in box)
    return max(x1, 0), max(y1, 0), min(x2, W), min(y2, H)  # numpy slicing semantics of img[y1:y2, x1:x2]


_BLOB_STAGING = __import__("threading").local()  # per thread: a ring of pinned int32 buffers, each with the event of its last copy
_BLOB_SLOTS = 8


def _stage_blob(words: np.ndarray, device) -> torch.Tensor:
    """int32 words -> device through a pinned buffer of the calling thread, asynchronously on the current stream: no pageable
    staging inside the runtime (a blocking copy in small chunks) and no wait for the stream.  A buffer is written again only
    after the copy that last read it has completed (its event; eight buffers in turn, so that is long past)."""
    st = getattr(_BLOB_STAGING, "slots", None)
    if st is None:
        st = _BLOB_STAGING.slots = [{"buf": None, "event": None} for _ in range(_BLOB_SLOTS)]
        _BLOB_STAGING.turn = 0
    slot = st[_BLOB_STAGING.turn]
    _BLOB_STAGING.turn = (_BLOB_STAGING.turn + 1) % _BLOB_SLOTS
    if slot["event"] is not None:
        slot["event"].synchronize()
    n = int(words.size)
    if slot["buf"] is None or slot["buf"].numel() < n:
        slot["buf"] = torch.empty(max(n, 1 << 16), dtype=torch.int32, pin_memory=True)
    host = slot["buf"][:n]
    np.copyto(host.numpy(), words)
    dev = torch.empty(n, dtype=torch.int32, device=device)
    dev.copy_(host, non_blocking=True)
    slot["event"] = torch.cuda.Event()
    slot["event"].record(torch.cuda.current_stream(device))
    return dev


_HOST_STAGING = __import__("threading").local()  # per thread: one pinned byte buffer for results on their way to numpy


def to_host(*tensors: torch.Tensor):
    """Device tensors -> numpy arrays (copies the caller owns) through ONE pinned buffer of the calling thread: every tensor is
    queued as one asynchronous DMA, the current stream is waited for once, and the arrays are copied out of the buffer.  The
    `.cpu()` of a device tensor lands in pageable memory, which the runtime fills through a staging buffer in small chunks
    - a blit launch and a wait per chunk."""
    if not tensors:
        return ()
    dev = tensors[0].device
    srcs = [t.detach().contiguous() for t in tensors]
    sizes = [(t.numel() * t.element_size() + 63) // 64 * 64 for t in srcs]
    total = sum(sizes)
    buf = getattr(_HOST_STAGING, "buf", None)
    if buf is None or buf.numel() < total:
        buf = _HOST_STAGING.buf = torch.empty(max(total, 1 << 20), dtype=torch.uint8, pin_memory=True)
    views, off = [], 0
    for t, nbytes in zip(srcs, sizes):
        view = buf[off : off + t.numel() * t.element_size()].view(t.dtype).view(t.shape)
        view.copy_(t, non_blocking=True)
        views.append(view)
        off += nbytes
    torch.cuda.current_stream(dev).synchronize()
    return tuple(v.numpy().copy() for v in views)


def rtdetr_batch_tensor(pages_dev: Sequence[torch.Tensor], crops: Sequence, out_hw=(640, 640), out: torch.Tensor = None):
    """`rtdetr_tensor` for many crops in ONE launch: crops = [(page index, box or None)] -> (fp32 n x 3 x H x W on the device,
    [{"size": (crop height, crop width), "offset": (x1, y1)}]).  The coefficient tables of all crops travel in one
    host-to-device copy (a pinned buffer of the calling thread, on the current stream); every crop's values are those of
    the one-crop form (tests/test_imaging_gpu.py)."""
    oh, ow = (int(v) for v in out_hw)
    n = len(crops)
    dev = pages_dev[0].device if len(pages_dev) else torch.device("cpu")
    if out is None:
        out = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=dev)
    if n == 0:
        return out, []
    lib = _lib.load()
    rec_words = lib.ymk_pil_batch_record_words()
    tables, index, metas = [], {}, []
    off = n * rec_words
    head = np.zeros((n, rec_words), dtype=np.int64)

    def table(in_size, out_size):
        nonlocal off
        hit = index.get((in_size, out_size))
        if hit is None:
            b, c, k = pil_bilinear_coeffs(in_size, out_size)
            hit = index[(in_size, out_size)] = (off, off + b.size, k)
            tables.append(b.reshape(-1))
            tables.append(c.reshape(-1))
            off += b.size + c.size
        return hit

    for i, (p, box) in enumerate(crops):
        page = pages_dev[p]
        x1, y1, x2, y2 = _clamped_box(page, box)
        cw, ch = x2 - x1, y2 - y1
        if cw <= 0 or ch <= 0:
            raise ValueError(f"empty crop {box}")
        xb, xk, ksx = table(cw, ow)
        yb, yk, ksy = table(ch, oh)
        ptr = int(page.data_ptr())
        head[i, :11] = (ptr & 0xFFFFFFFF, ptr >> 32, int(page.shape[1]), x1, y1, ksx, ksy, xb, xk, yb, yk)
        metas.append({"size": (ch, cw), "offset": (x1, y1)})
    words = np.concatenate([head.astype(np.uint32).view(np.int32).reshape(-1)] + tables)
    blob = _stage_blob(words, dev)
    if tuple(out.shape) != (n, 3, oh, ow) or not out.is_contiguous():
        raise ValueError(f"rtdetr_batch_tensor: out must be a contiguous {n} x 3 x {oh} x {ow} tensor")
    with torch.cuda.device(dev):
        _lib.check(lib.ymk_pil_resize_batch_to_chw(blob.data_ptr(), n, oh, ow, out.data_ptr(), _lib.current_stream_ptr()),
                   "ymk_pil_resize_batch_to_chw")
    return out, metas


# ---------------------------------------------------------------------------------------------
# recogniser crops
class CropDesc(ctypes.Structure):
    _fields_ = [
        ("minv", ctypes.c_double * 9),
        ("bx", ctypes.c_int), ("by", ctypes.c_int), ("bw", ctypes.c_int), ("bh", ctypes.c_int),
        ("ww", ctypes.c_int), ("wh", ctypes.c_int), ("rot", ctypes.c_int),
        ("rw", ctypes.c_int), ("rh", ctypes.c_int), ("nw", ctypes.c_int), ("nh", ctypes.c_int),
        ("fast_x", ctypes.c_int), ("fast_y", ctypes.c_int),
        ("warp_off", ctypes.c_longlong),
        ("slot", ctypes.c_int),
        ("flip", ctypes.c_int),
        ("level", ctypes.c_int),
    ]


@dataclass
class CropPlan:
    index: int              # position in the caller's quad list
    desc: CropDesc
    content_width: int      # calc_resize_without_padding width (batch bucketing key)
    canvas_width: int       # width of this crop's own tensor (800, or the dynamic canvas)
    table: Optional[np.ndarray] = None  # the page's descriptor records (`desc` is a view of row `row`), for bulk gathers
    row: int = 0


def validate_quad(shape_hw, quad) -> bool:
    """validate_quads (data/functions.py:267-298): 4 points of 2, bbox inside the image."""
    if len(quad) != 4 or any(len(p) != 2 for p in quad):
        return False
    q = np.array(quad, dtype=int)
    h, w = shape_hw
    return not (q[:, 0].min() < 0 or q[:, 0].max() > w or q[:, 1].min() < 0 or q[:, 1].max() > h)


def perspective_matrix(src4: np.ndarray, dst4: np.ndarray) -> np.ndarray:
    """cv2.getPerspectiveTransform: 8 x 8 linear solve in double."""
    a = np.zeros((8, 8), dtype=np.float64)
    b = np.zeros(8, dtype=np.float64)
    for i in range(4):
        x, y = float(src4[i][0]), float(src4[i][1])
        u, v = float(dst4[i][0]), float(dst4[i][1])
        a[i, 0], a[i, 1], a[i, 2] = x, y, 1.0
        a[i + 4, 3], a[i + 4, 4], a[i + 4, 5] = x, y, 1.0
        a[i, 6], a[i, 7] = -x * u, -y * u
        a[i + 4, 6], a[i + 4, 7] = -x * v, -y * v
        b[i], b[i + 4] = u, v
    return np.append(np.linalg.solve(a, b), 1.0).reshape(3, 3)


def _int_scale(src: int, dst: int) -> int:
    s = src / dst
    i = round(s)
    return i if abs(s - i) < 2.220446049250313e-16 else 0


_CROP_DTYPE = np.dtype(CropDesc)  # the ctypes layout as a numpy record: descriptors are filled column-wise


def plan_crops(shape_hw, quads, img_size=(32, 800), dynamic_width=False, align=8, margin=64) -> List[Optional[CropPlan]]:
    """Integer geometry of every text-line crop (extract_roi_with_perspective, rotate_text_image,
    calc_resize_without_padding, resize_with[_dynamic]_padding).  None for quads that fail validation.
    All quads of the page at once: one stacked 8 x 8 solve + 3 x 3 inverse (LAPACK per matrix, so the
    doubles are those of the per-quad form, `_plan_crops_scalar`)."""
    n = len(quads)
    plans: List[Optional[CropPlan]] = [None] * n
    shaped = [i for i, quad in enumerate(quads) if len(quad) == 4 and all(len(p) == 2 for p in quad)]
    if not shaped:
        return plans
    th, tw = int(img_size[0]), int(img_size[1])
    q = np.array([quads[i] for i in shaped], dtype=np.int64)  # m x 4 x 2, truncated like np.array(quad, dtype=int)
    h, w = shape_hw
    bx, by = q[:, :, 0].min(1), q[:, :, 1].min(1)
    bx2, by2 = q[:, :, 0].max(1), q[:, :, 1].max(1)
    inside = ~((bx < 0) | (bx2 > w) | (by < 0) | (by2 > h))
    if not inside.all():
        shaped = [i for i, ok in zip(shaped, inside.tolist()) if ok]
        q, bx, by, bx2, by2 = q[inside], bx[inside], by[inside], bx2[inside], by2[inside]
        if not shaped:
            return plans
    m = len(shaped)
    rel = q - np.stack([bx, by], axis=1)[:, None, :]
    width = np.sqrt(((rel[:, 0] - rel[:, 1]) ** 2).sum(1).astype(np.float64)).astype(np.int64)
    height = np.sqrt(((rel[:, 1] - rel[:, 2]) ** 2).sum(1).astype(np.float64)).astype(np.int64)
    ok = (width > 0) & (height > 0)
    if not ok.all():
        # a quad whose first or second edge is shorter than one pixel: the reference hands OpenCV an empty dsize there
        # (the output is then whatever warpPerspective makes of a singular transform); here such a quad is dropped like
        # one that fails validate_quads, instead of aborting the page
        shaped = [i for i, keep in zip(shaped, ok.tolist()) if keep]
        q, bx, by, bx2, by2, rel, width, height = q[ok], bx[ok], by[ok], bx2[ok], by2[ok], rel[ok], width[ok], height[ok]
        m = len(shaped)
        if m == 0:
            return plans
    # cv2.getPerspectiveTransform(rel -> [[0,0],[w,0],[w,h],[0,h]]) for every quad
    x, y = rel[:, :, 0].astype(np.float64), rel[:, :, 1].astype(np.float64)
    wf, hf = width.astype(np.float64), height.astype(np.float64)
    zero = np.zeros(m)
    u, v = np.stack([zero, wf, wf, zero], 1), np.stack([zero, zero, hf, hf], 1)
    a = np.zeros((m, 8, 8), dtype=np.float64)
    a[:, :4, 0], a[:, :4, 1], a[:, :4, 2] = x, y, 1.0
    a[:, 4:, 3], a[:, 4:, 4], a[:, 4:, 5] = x, y, 1.0
    a[:, :4, 6], a[:, :4, 7] = -x * u, -y * u
    a[:, 4:, 6], a[:, 4:, 7] = -x * v, -y * v
    sol = np.linalg.solve(a, np.concatenate([u, v], 1)[:, :, None])[:, :, 0]
    minv = np.linalg.inv(np.concatenate([sol, np.ones((m, 1))], 1).reshape(m, 3, 3))
    rot = height > 2 * width
    rw, rh = np.where(rot, height, width), np.where(rot, width, height)
    s = np.minimum(np.where(rw > tw, tw / rw, 1.0), np.where(rh > th, th / rh, 1.0))
    nw = np.maximum(1, (rw * s).astype(np.int64))
    nh = np.maximum(1, (rh * s).astype(np.int64))
    canvas = np.minimum(tw, ((nw + margin + align - 1) // align) * align) if dynamic_width else np.full(m, tw)

    def int_scale(src, dst):
        sc = src / dst
        r = np.rint(sc)
        return np.where(np.abs(sc - r) < 2.220446049250313e-16, r, 0).astype(np.int64)

    fx, fy = int_scale(rw, nw), int_scale(rh, nh)
    fast = (fx != 0) & (fy != 0)
    rec = np.zeros(m, dtype=_CROP_DTYPE)
    rec["minv"] = minv.reshape(m, 9)
    rec["bx"], rec["by"], rec["bw"], rec["bh"] = bx, by, bx2 - bx, by2 - by
    rec["ww"], rec["wh"], rec["rot"], rec["rw"], rec["rh"], rec["nw"], rec["nh"] = width, height, rot, rw, rh, nw, nh
    rec["fast_x"], rec["fast_y"] = np.where(fast, fx, 0), np.where(fast, fy, 0)
    descs = (CropDesc * m).from_buffer(rec)  # views into `rec`; each element keeps the buffer alive
    for k, (i, cw, cv) in enumerate(zip(shaped, nw.tolist(), canvas.tolist())):
        plans[i] = CropPlan(index=i, desc=descs[k], content_width=cw, canvas_width=cv, table=rec, row=k)
    return plans


def _plan_crops_scalar(shape_hw, quads, img_size=(32, 800), dynamic_width=False, align=8, margin=64) -> List[Optional[CropPlan]]:
    """One quad at a time, statement for statement after the reference's helpers; the batched form above
    is tested against it (tests/test_imaging_host.py)."""
    plans: List[Optional[CropPlan]] = []
    th, tw = int(img_size[0]), int(img_size[1])
    for i, quad in enumerate(quads):
        if not validate_quad(shape_hw, quad):
            plans.append(None)
            continue
        q = np.array(quad, dtype=np.int64)
        bx, by = int(q[:, 0].min()), int(q[:, 1].min())
        bx2, by2 = int(q[:, 0].max()), int(q[:, 1].max())
        rel = q.copy()
        rel[:, 0] -= bx
        rel[:, 1] -= by
        width = int(np.linalg.norm(rel[0] - rel[1]))
        height = int(np.linalg.norm(rel[1] - rel[2]))
        if width <= 0 or height <= 0:  # dropped, like a quad that fails validation (see plan_crops)
            plans.append(None)
            continue
        M = perspective_matrix(np.float32(rel), np.float32([[0, 0], [width, 0], [width, height], [0, height]]))
        minv = np.linalg.inv(M)
        rot = 1 if height > 2 * width else 0
        rw, rh = (height, width) if rot else (width, height)
        scale_w = tw / rw if rw > tw else 1.0
        scale_h = th / rh if rh > th else 1.0
        s = min(scale_w, scale_h)
        nw, nh = max(1, int(rw * s)), max(1, int(rh * s))
        canvas = min(tw, ((nw + margin + align - 1) // align) * align) if dynamic_width else tw
        d = CropDesc()
        for k in range(9):
            d.minv[k] = float(minv.flat[k])
        d.bx, d.by, d.bw, d.bh = bx, by, bx2 - bx, by2 - by
        d.ww, d.wh, d.rot, d.rw, d.rh, d.nw, d.nh = width, height, rot, rw, rh, nw, nh
        fx, fy = _int_scale(rw, nw), _int_scale(rh, nh)
        d.fast_x, d.fast_y = (fx, fy) if fx and fy else (0, 0)
        plans.append(CropPlan(index=i, desc=d, content_width=nw, canvas_width=canvas))
    return plans


def plan_crops_pyramid(shape_hw, quads, img_size=(32, 800), dynamic_width=False, source_downscale=False):
    """plan_crops with the reference's source_downscale routing (data/dataset.py:64-103): each quad is planned on its
    pyramid level, with its corners divided by 2^level in float32 (then truncated like any quad).  Returns
    (plans, levels): plans[i].desc.level names the level; coordinates reported to the caller stay in page space."""
    n = len(quads)
    levels = source_levels(quads, img_size[0]) if source_downscale and n > 0 else np.zeros(n, dtype=int)
    if not levels.any():
        return plan_crops(shape_hw, quads, img_size, dynamic_width), levels
    shapes = [tuple(int(v) for v in shape_hw[:2])]
    for _ in range(int(levels.max())):
        shapes.append((half_size(shapes[-1][0]), half_size(shapes[-1][1])))
    plans: List[Optional[CropPlan]] = [None] * n
    for k in sorted(set(levels.tolist())):
        idx = np.flatnonzero(levels == k).tolist()
        sub_quads = [quads[i] if k == 0 else (np.asarray(quads[i], dtype=np.float32) / (2.0 ** k)).tolist() for i in idx]
        for i, plan in zip(idx, plan_crops(shapes[k], sub_quads, img_size, dynamic_width)):
            if plan is not None:
                plan.index = i
                plan.desc.level = k
                plans[i] = plan
    return plans, levels


def source_levels(quads, target_height: int, max_level: int = 3) -> np.ndarray:
    """_calc_source_levels (data/dataset.py:16-41): a crop whose short side is s can be cut from the 2^k-downscaled
    page as long as s / 2^k >= target_height; k = clip(floor(log2(s / target_height)), 0, max_level) per quad."""
    if len(quads) == 0:
        return np.zeros(0, dtype=int)
    q = np.asarray(quads, dtype=np.float32).reshape(-1, 4, 2)
    w = np.linalg.norm(q[:, 0] - q[:, 1], axis=1)
    h = np.linalg.norm(q[:, 1] - q[:, 2], axis=1)
    short = np.maximum(1.0, np.minimum(w, h))
    return np.clip(np.floor(np.log2(short / float(target_height))).astype(int), 0, max_level)


def half_size(n: int) -> int:
    """cv2.resize(..., fx=0.5): dsize = saturate_cast<int>(n * 0.5) = round half to even."""
    return int(round(n * 0.5))


def build_pyramid(page_dev: torch.Tensor, needed_levels) -> List[Optional[torch.Tensor]]:
    """Levels 0..max(needed) of the page as uint8 H x W x 3 device tensors (2x INTER_AREA halvings, each from the
    previous level, data/dataset.py:73-79); levels no quad uses are built only as stepping stones and dropped."""
    lib = _lib.load()
    needed = {int(v) for v in needed_levels}
    top = max(needed) if needed else 0
    levels: List[Optional[torch.Tensor]] = [page_dev]
    cur = page_dev
    for k in range(1, top + 1):
        h, w = cur.shape[:2]
        nxt = torch.empty((half_size(h), half_size(w), 3), dtype=torch.uint8, device=page_dev.device)
        with torch.cuda.device(page_dev.device):
            _lib.check(lib.ymk_halve_u8c3(cur.data_ptr(), h, w, nxt.data_ptr(), nxt.shape[0], nxt.shape[1], _lib.current_stream_ptr()),
                       "ymk_halve_u8c3")
        levels.append(nxt if any(v >= k for v in needed) else None)
        cur = nxt
    return levels


def build_crop_batch(page_dev, plans: Sequence[CropPlan], out_h: int = 32, batch_w: Optional[int] = None, flip: bool = False):
    """Run the warp + resize kernels for one mini-batch: fp32 B x 3 x out_h x batch_w on the device.
    page_dev: the resident page, or the list of pyramid levels (`build_pyramid`) when plans carry levels.
    flip: every crop is turned by 180 degrees before the resize (orientation-fallback retry)."""
    lib = _lib.load()
    assert ctypes.sizeof(CropDesc) == lib.ymk_crop_desc_size(), "CropDesc layout mismatch"
    levels = list(page_dev) if isinstance(page_dev, (list, tuple)) else [page_dev]
    page0 = levels[0]
    n = len(plans)
    if batch_w is None:
        batch_w = max(p.canvas_width for p in plans)
    # the mini-batch's descriptor records, gathered per source table (one per pyramid level) instead of plan by plan
    arr = np.empty(n, dtype=_CROP_DTYPE)
    by_table = {}
    for slot, p in enumerate(plans):
        if p.table is None:  # a stand-alone descriptor (scalar planner)
            arr[slot] = np.frombuffer(p.desc, dtype=_CROP_DTYPE, count=1)[0]
        else:
            by_table.setdefault(id(p.table), (p.table, [], []))
            by_table[id(p.table)][1].append(slot)
            by_table[id(p.table)][2].append(p.row)
    for table, slots, rows in by_table.values():
        arr[slots] = table[rows]
    lv = arr["level"]
    if int(lv.max()) >= len(levels) or any(levels[int(k)] is None for k in np.unique(lv)):
        raise ValueError(f"crop plan wants pyramid level {int(lv.max())}, which was not built")
    arr["slot"] = np.arange(n)
    arr["flip"] = 1 if flip else 0
    sizes = (arr["ww"].astype(np.int64) * arr["wh"] * 3 + 15) & ~15
    ends = np.cumsum(sizes)
    arr["warp_off"] = ends - sizes
    off = int(ends[-1])
    max_w, max_h = max(1, int(arr["ww"].max())), max(1, int(arr["wh"].max()))
    dev = page0.device
    descs = _stage_blob(arr.view(np.int32).reshape(-1), dev)  # one asynchronous upload from pinned memory (was: a blocking pageable copy per mini-batch)
    scratch = torch.empty(max(off, 16), dtype=torch.uint8, device=dev)
    out = torch.empty((n, 3, out_h, batch_w), dtype=torch.float32, device=dev)
    nl = len(levels)
    ptrs = (ctypes.c_void_p * nl)(*[(t.data_ptr() if t is not None else None) for t in levels])
    hs = (ctypes.c_int * nl)(*[(t.shape[0] if t is not None else 0) for t in levels])
    ws = (ctypes.c_int * nl)(*[(t.shape[1] if t is not None else 0) for t in levels])
    with torch.cuda.device(dev):
        _lib.check(
            lib.ymk_crop_batch_levels(ptrs, hs, ws, nl, descs.data_ptr(), n, max_w, max_h, scratch.data_ptr(), out.data_ptr(),
                                      batch_w, out_h, _lib.current_stream_ptr()),
            "ymk_crop_batch_levels",
        )
    return out

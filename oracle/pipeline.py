"""ORACLE (test infrastructure only - never imported by the product path).

CPU restatement of the module-level call chains of the reference (text_detector.py,
text_recognizer.py, layout_parser.py, table_structure_recognizer.py and
postprocessor/rtdetr_postprocessor.py), composed from the oracle nets and oracle/cvlike.py.  Used
(a) stage by stage in tests/test_pipeline_gpu.py and (b) as the `cpu_baseline` leg of bench.py
(kind "port": the reference itself cannot be imported - cv2 / torchvision / timm / omegaconf are
not installed - see BASELINE.md §4).
"""

from __future__ import annotations

import unicodedata

import numpy as np
import torch

from . import cvlike, preprocess
from .dbnet import dbnet_forward
from .parseq import parseq_forward, tokenizer_decode
from .rtdetr import rtdetr_forward


# ------------------------------------------------------------------ TextDetector.__call__ (text_detector.py:112-146)
def detect(sd, img_bgr, shortest=1280, limit=1600, post=None, prob=None):
    post = post or dict(min_size=2, thresh=0.3, box_thresh=0.4, max_candidates=1500, unclip_ratio=3.5)
    if prob is None:
        prob = dbnet_forward(sd, preprocess.detector_preprocess(img_bgr, shortest, limit))["binary"]
    pred = prob[0, 0].numpy() if isinstance(prob, torch.Tensor) else prob[0, 0]
    quads, scores = cvlike.db_postprocess(pred, img_bgr.shape[:2], post["min_size"], post["thresh"], post["box_thresh"],
                                          post["max_candidates"], post["unclip_ratio"])
    return prob, quads, scores


# ------------------------------------------------------------------ TextRecognizer.__call__ (text_recognizer.py:115-399)
def make_batches(widths_canvas, order, dynamic_width, width_budget, max_batch_size, batch_size):
    indices = order if order is not None else range(len(widths_canvas))
    batches, cur = [], []
    if dynamic_width and width_budget:
        cur_max = 0
        for idx in indices:
            w = widths_canvas[idx]
            new_max = w if w > cur_max else cur_max
            if cur and ((len(cur) + 1) * new_max > width_budget or (max_batch_size is not None and len(cur) >= max_batch_size)):
                batches.append(cur)
                cur = []
                new_max = w
            cur.append(idx)
            cur_max = new_max
        if cur:
            batches.append(cur)
        return batches
    for idx in indices:
        cur.append(idx)
        if len(cur) == batch_size:
            batches.append(cur)
            cur = []
    if cur:
        batches.append(cur)
    return batches


def recognize(sd, ocfg, img_bgr, quads, charset, dynamic_width=False, batch_bucketing=False, width_budget=None,
              max_batch_size=None, batch_size=128, forward=None, source_downscale=False, orientation_fallback=False,
              fallback_thresh=0.75):
    """-> (contents, scores, directions) in detection order (TextRecognizer.__call__, text_recognizer.py:352-399, with
    ParseqDataset's source_downscale routing, data/dataset.py:64-103, and the 180-degree retry, :319-350)."""
    levels = {0: img_bgr[:, :, ::-1]}
    quad_levels = np.zeros(len(quads), dtype=int)
    if source_downscale and len(quads) > 0:
        quad_levels = preprocess.calc_source_levels(quads, ocfg.img_size[0])
        level_img = np.ascontiguousarray(img_bgr)
        for k in range(1, int(quad_levels.max()) + 1):
            level_img = cvlike.resize_half(level_img)
            if (quad_levels >= k).any():
                levels[k] = level_img[:, :, ::-1]
    crops = []
    for q, k in zip(quads, quad_levels):
        qq = (np.asarray(q, dtype=np.float32) / (2.0 ** int(k))).tolist() if k > 0 else q
        crops.append(preprocess.parseq_crop(levels.get(int(k), levels[0]), qq, ocfg.img_size, dynamic_width, with_roi=True))
    data = [c for c in crops if c is not None]
    tensors = [c[0] for c in data]
    widths = [c[1] for c in data]
    order = None
    if batch_bucketing and len(data) == len(quads) and len(data) > 1:
        order = np.argsort(widths).tolist()
    batches = make_batches([t.shape[-1] for t in tensors], order, dynamic_width, width_budget, max_batch_size, batch_size)
    itos = ("[E]",) + tuple(charset) + ("[B]", "[P]")
    pts = [quads[i] for i in order] if order is not None else quads
    preds, scores, directions = [], [], []
    off = 0
    forward = forward or (lambda x: parseq_forward(sd, ocfg, x))
    for b in batches:
        ts = [tensors[i] for i in b]
        if dynamic_width:
            mw = max(t.shape[-1] for t in ts)
            ts = [torch.nn.functional.pad(t, (0, mw - t.shape[-1]), value=-1.0) for t in ts]
        p = forward(torch.stack(ts, 0)).softmax(-1)
        ids, sc = tokenizer_decode(p)
        preds += [unicodedata.normalize("NFKC", "".join(itos[i] for i in row)) for row in ids]
        scores += sc
        for point in pts[off : off + len(b)]:
            point = np.array(point)
            w, h = np.linalg.norm(point[0] - point[1]), np.linalg.norm(point[1] - point[2])
            directions.append("vertical" if h > w * 2 else "horizontal")
        off += len(b)
    if order is not None:
        inv = np.argsort(order)
        preds, scores, directions = [preds[i] for i in inv], [scores[i] for i in inv], [directions[i] for i in inv]
    if orientation_fallback:
        retry = [i for i, s_ in enumerate(scores) if s_ < fallback_thresh]
        if retry:
            flipped = [preprocess.canvas_tensor(np.ascontiguousarray(np.rot90(data[i][2], 2)), ocfg.img_size)[0] for i in retry]
            r_preds, r_scores, r_dirs = [], [], []
            for s0 in range(0, len(retry), batch_size):
                p = forward(torch.stack(flipped[s0 : s0 + batch_size], 0)).softmax(-1)
                ids, sc = tokenizer_decode(p)
                r_preds += [unicodedata.normalize("NFKC", "".join(itos[i] for i in row)) for row in ids]
                r_scores += sc
                for i in retry[s0 : s0 + batch_size]:
                    point = np.array(quads[i])
                    w, h = np.linalg.norm(point[0] - point[1]), np.linalg.norm(point[1] - point[2])
                    r_dirs.append("vertical" if h > w * 2 else "horizontal")
            for j, i in enumerate(retry):
                if r_scores[j] > scores[i] and r_scores[j] >= fallback_thresh:
                    preds[i], scores[i], directions[i] = r_preds[j], r_scores[j], r_dirs[j]
    return preds, scores, directions


# ------------------------------------------------------------------ RTDETRPostProcessor.forward (rtdetr_postprocessor.py:60-123)
def rtdetr_post(logits, boxes, orig_wh, threshold, num_classes, num_top_queries=300):
    logits, boxes = torch.as_tensor(logits), torch.as_tensor(boxes)
    cx, cy, w, h = boxes.unbind(-1)
    bbox = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)  # torchvision box_convert
    size = torch.tensor([orig_wh])
    bbox = bbox * size.repeat(1, 2).unsqueeze(1)
    scores = torch.sigmoid(logits)
    scores, index = torch.topk(scores.flatten(1), num_top_queries, dim=-1)
    labels = index - index // num_classes * num_classes
    index = index // num_classes
    bsel = bbox.gather(dim=1, index=index.unsqueeze(-1).repeat(1, 1, 4))
    out = []
    for lab, box, sco in zip(labels, bsel, scores):
        keep = sco > threshold
        lab, box, sco = lab[keep].numpy(), box[keep].clone(), sco[keep].numpy()
        box[:, 0] = torch.clamp(box[:, 0], min=0)
        box[:, 1] = torch.clamp(box[:, 1], min=0)
        box[:, 2] = torch.clamp(box[:, 2], min=0, max=float(orig_wh[0]))
        box[:, 3] = torch.clamp(box[:, 3], min=0, max=float(orig_wh[1]))
        out.append(dict(labels=lab, boxes=box.numpy(), scores=sco))
    return out


def layout(sd, img_bgr, thresh=0.5, nc=6, forward=None):
    x, _ = preprocess.rtdetr_preprocess(img_bgr)
    preds = (forward or (lambda t: rtdetr_forward(sd, t)))(x)
    h, w = img_bgr.shape[:2]
    return preds, rtdetr_post(preds["pred_logits"], preds["pred_boxes"], (w, h), thresh, nc)[0]


def tables(sd, img_bgr, table_boxes, thresh=0.4, nc=3, forward=None):
    out = []
    for box in table_boxes:
        x, (th, tw) = preprocess.rtdetr_preprocess(img_bgr, box)
        preds = (forward or (lambda t: rtdetr_forward(sd, t)))(x)
        out.append((preds, rtdetr_post(preds["pred_logits"], preds["pred_boxes"], (tw, th), thresh, nc)[0]))
    return out


# ------------------------------------------------------------------ DocumentAnalyzer.__call__ (document_analyzer.py:622-678), free-running
def analyze(sds, ocfg, img_bgr, charset, rec_opts=None, det_opts=None, agg_opts=None, forwards=None, keep=None, split_text_across_cells=False):
    """The whole page on the CPU with nothing taken from the product: detector -> boxes -> recogniser ‖ layout -> table crops
    -> table structure -> aggregation and reading order (oracle.hostlogic, pinned against the reference's own functions).
    `split_text_across_cells`: the option of the same name (the quads are cut at cell borders before they are recognised).
    sds: state dicts {"det", "rec", "lay", "tab"}; returns the page record as plain dicts, shaped like
    DocumentAnalyzerSchema.model_dump().  `forwards`: optional replacements for the four network forwards (timing legs);
    `keep`: a dict that receives the continuous stage outputs (probability map, layout / table logits and boxes, the
    thresholded detections) for margin analysis (tools/e2e_oracle_eval.py)."""
    from . import hostlogic as hl

    rec_opts = dict(dynamic_width=True, batch_bucketing=True, width_budget=8000, max_batch_size=64, batch_size=10,
                    source_downscale=True) if rec_opts is None else rec_opts
    forwards = forwards or {}
    prob, quads, det_scores = detect(sds["det"], img_bgr, **(det_opts or {}))
    lay_preds, lay_det = layout(sds["lay"], img_bgr, forward=forwards.get("lay"))
    groups = hl.layout_elements(lay_det)
    table_boxes = [t["box"] for t in groups["tables"]]
    structures, tab_raw = [], []
    for box, (preds, det) in zip(table_boxes, tables(sds["tab"], img_bgr, table_boxes, forward=forwards.get("tab"))):
        x1, y1, x2, y2 = (int(v) for v in box)
        th, tw = img_bgr[y1:y2, x1:x2].shape[:2]
        table = hl.table_structure(det, (th, tw), (x1, y1))
        tab_raw.append((preds, det, table))
        if table["n_row"] > 0 and table["n_col"] > 0:
            structures.append(table)
    if split_text_across_cells:  # document_analyzer.py:636-655: the detector's quads are cut at the cell borders BEFORE recognition
        quads, det_scores = hl.split_text_across_cells([[[int(x), int(y)] for x, y in q] for q in quads], list(det_scores), structures)
    contents, rec_scores, directions = recognize(sds["rec"], ocfg, img_bgr, quads, charset, forward=forwards.get("rec"), **rec_opts)
    words = [{"points": [[int(x), int(y)] for x, y in q], "content": c, "direction": d, "rec_score": float(rs), "det_score": float(ds)}
             for q, ds, c, rs, d in zip(quads, det_scores, contents, rec_scores, directions)]
    if keep is not None:
        keep.update(prob=prob, quads=quads, det_scores=det_scores, rec_scores=rec_scores, lay_preds=lay_preds, lay_det=lay_det,
                    layout_groups={k: [dict(e) for e in v] for k, v in groups.items()}, tab_raw=tab_raw)
    return hl.aggregate(words, {"paragraphs": groups["paragraphs"], "tables": structures, "figures": groups["figures"]}, **(agg_opts or {}))

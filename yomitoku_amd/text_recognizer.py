"""TextRecognizer module (reference text_recognizer.py:35-399): same constructor, catalog names,
config keys, batching rules and `__call__(img, points, vis) -> (TextRecognizerSchema, vis)`.

The batching rules are numerics, not plumbing (padding columns are ordinary ViT
tokens), so `np.argsort` bucketing and the width-budget rule are reproduced literally; what moves
to the MI355X is the crop warp / resize / normalise / pad (one fused kernel pair per mini-batch),
the PARSeq forward, and the softmax -> (arg-max, max-prob) reduction the tokenizer needs."""

from __future__ import annotations

import logging
import threading
import unicodedata
from typing import List

import numpy as np
import torch

from . import imaging
from .base import BaseModelCatalog, BaseModule
from .configs import (
    TextRecognizerPARSeqConfig,
    TextRecognizerPARSeqLargeV41Config,
    TextRecognizerPARSeqSmallConfig,
    TextRecognizerPARSeqTinyConfig,
    TextRecognizerPARSeqTinyDynwV4Config,
    TextRecognizerPARSeqV2Config,
)
from .nets import PARSeq
from .schemas import TextRecognizerSchema


logger = logging.getLogger(__name__)


def load_charset(charset_path):
    with open(charset_path, "r", encoding="utf-8") as f:
        return f.read()


class ParseqTokenizer:
    """postprocessor/parseq_tokenizer.py:91-126: itos = [E] + charset + [B], [P]."""

    BOS, EOS, PAD = "[B]", "[E]", "[P]"

    def __init__(self, charset: str):
        self._itos = (self.EOS,) + tuple(charset) + (self.BOS, self.PAD)
        self._stoi = {s: i for i, s in enumerate(self._itos)}
        self._chars = np.array(self._itos, dtype=object)
        self.eos_id, self.bos_id, self.pad_id = (self._stoi[s] for s in (self.EOS, self.BOS, self.PAD))

    def __len__(self):
        return len(self._itos)

    def decode_stats(self, ids: np.ndarray, probs: np.ndarray):
        """Greedy decode from per-position (arg-max id, max prob): cut at the first <eos>, the score is
        the product of the kept probabilities including the <eos> one (parseq_tokenizer.py:64-88,117-126)."""
        ids = np.asarray(ids)
        if ids.ndim != 2 or ids.shape[0] == 0:
            return [], []
        n, s = ids.shape
        is_eos = ids == self.eos_id
        end = np.where(is_eos.any(1), is_eos.argmax(1), s)  # tokens kept per row
        # float32 running product, left to right = np.prod of the float32 slice row_p[: e + 1]
        run = np.cumprod(np.asarray(probs, dtype=np.float32), axis=1, dtype=np.float32)
        scores = run[np.arange(n), np.minimum(end, s - 1)].astype(np.float64).tolist()
        chars = self._chars
        texts = ["".join(chars[row[:e]].tolist()) for row, e in zip(ids, end.tolist())]
        return texts, scores

    def decode(self, token_dists):
        """API parity: softmax probabilities N x L x C (torch) -> (labels, scores)."""
        p, i = token_dists.max(-1)
        return self.decode_stats(i.cpu().numpy(), p.cpu().numpy())


class TextRecognizerModelCatalog(BaseModelCatalog):
    def __init__(self):
        super().__init__()
        self.register("parseq", TextRecognizerPARSeqConfig, PARSeq)
        self.register("parseqv2", TextRecognizerPARSeqV2Config, PARSeq)
        self.register("parseq-small", TextRecognizerPARSeqSmallConfig, PARSeq)
        self.register("parseq-tiny", TextRecognizerPARSeqTinyConfig, PARSeq)
        self.register("parseq-large-v4_1", TextRecognizerPARSeqLargeV41Config, PARSeq)
        self.register("parseq-tiny-dynw-v4", TextRecognizerPARSeqTinyDynwV4Config, PARSeq)


class CropSet:
    """What ParseqDataset holds (data/dataset.py:44-129), minus the pixels: the per-quad plans."""

    def __init__(self, cfg, page_dev, quads, dynamic_width=False, source_downscale=False):
        plans, levels = imaging.plan_crops_pyramid(page_dev.shape[:2], quads, cfg.data.img_size, dynamic_width, source_downscale)
        # the page, or its pyramid when some quads are cut from a 2^k-downscaled copy (dataset.py:64-79)
        self.page = imaging.build_pyramid(page_dev, levels) if levels.any() else page_dev
        self.plans = [p for p in plans if p is not None]
        self.content_widths = [p.content_width for p in self.plans]
        self.valid_quads = [q for q, p in zip(quads, plans) if p is not None]
        # quads without a plan: "invalid" fails validate_quads (the reference drops those from the dataset but not from
        # `points`, data/dataset.py:91-95 - kept as is); "degenerate" passes it but has an edge shorter than one pixel
        # (the reference hands OpenCV an empty dsize there) - those get a placeholder result so that contents / scores /
        # directions stay aligned with `points`
        self.slots = ["ok" if p is not None else ("degenerate" if imaging.validate_quad(page_dev.shape[:2], q) else "invalid")
                      for q, p in zip(quads, plans)]
        self.n_degenerate = self.slots.count("degenerate")
        if len(self.plans) != len(plans):
            logger.warning("text recogniser: %d of %d quads dropped (%d outside the page or malformed, %d with an edge shorter "
                           "than one pixel - the latter are reported as empty strings with score 0)", len(plans) - len(self.plans),
                           len(plans), self.slots.count("invalid"), self.n_degenerate)

    def __len__(self):
        return len(self.plans)


class TextRecognizer(BaseModule):
    model_catalog = TextRecognizerModelCatalog()

    def __init__(self, model_name="parseq-large-v4_1", path_cfg=None, device="cuda", visualize=False, from_pretrained=True,
                 infer_onnx=False, rec_orientation_fallback=False, rec_orientation_fallback_thresh=0.75,
                 batch_bucketing=False, dynamic_width=False, num_parallel_batches=1, source_downscale=False):
        super().__init__()
        if infer_onnx:
            raise NotImplementedError("the ONNX backend is out of scope of the MI355X path (infer_onnx=False only)")
        self.load_model(model_name, path_cfg, from_pretrained=from_pretrained)
        self.charset = load_charset(self._cfg.charset)
        self.tokenizer = ParseqTokenizer(self.charset)
        if len(self.tokenizer) != int(self._cfg.num_tokens):
            raise ValueError(f"charset has {len(self.tokenizer)} tokens but the config says {self._cfg.num_tokens}")
        self.device = device
        self.model.tokenizer = self.tokenizer
        self.model.eval()
        self.visualize = visualize
        self.infer_onnx = False
        self.rec_orientation_fallback = bool(rec_orientation_fallback)
        self.rec_orientation_fallback_thresh = rec_orientation_fallback_thresh
        self.batch_bucketing = batch_bucketing
        self.dynamic_width = dynamic_width
        # text_recognizer.py:285-317 gates CPU worker threads on this (dead code there); here every
        # mini-batch of a call already shares one grouped forward, so the value is accepted and has no effect
        self.num_parallel_batches = int(num_parallel_batches)
        self.source_downscale = bool(source_downscale)
        self._replicas, self._replica_lock = {}, threading.Lock()
        self.model.to(self.device)

    # ------------------------------------------------------------------ batching (text_recognizer.py:115-203)
    def preprocess(self, img, polygons):
        page = img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device)
        if polygons is None:
            h, w = page.shape[:2]
            polygons = [[[0, 0], [w, 0], [w, h], [0, h]]]
        dataset = CropSet(self._cfg, page, polygons, dynamic_width=self.dynamic_width, source_downscale=self.source_downscale)
        order = None
        if self.batch_bucketing and len(dataset) == len(polygons) and len(dataset) > 1:
            order = np.argsort(dataset.content_widths).tolist()
        return self._make_mini_batch(dataset, order), polygons, dataset, order

    def _make_mini_batch(self, dataset, order=None) -> List[List[imaging.CropPlan]]:
        indices = order if order is not None else range(len(dataset))
        width_budget = getattr(self._cfg.data, "width_budget", None)
        batches: List[List[imaging.CropPlan]] = []
        cur: List[imaging.CropPlan] = []
        if self.dynamic_width and width_budget:
            max_batch_size = getattr(self._cfg.data, "max_batch_size", None)
            cur_max = 0
            for idx in indices:
                plan = dataset.plans[idx]
                w = plan.canvas_width
                new_max = w if w > cur_max else cur_max
                over_budget = (len(cur) + 1) * new_max > width_budget
                over_count = max_batch_size is not None and len(cur) >= max_batch_size
                if cur and (over_budget or over_count):
                    batches.append(cur)
                    cur = []
                    new_max = w
                cur.append(plan)
                cur_max = new_max
            if cur:
                batches.append(cur)
            return batches
        for idx in indices:
            cur.append(dataset.plans[idx])
            if len(cur) == self._cfg.data.batch_size:
                batches.append(cur)
                cur = []
        if cur:
            batches.append(cur)
        return batches

    def _collate(self, dataset, plans) -> torch.Tensor:
        """Crop kernels write the mini-batch tensor directly: B x 3 x 32 x W, W = widest canvas in the
        batch (dynamic width, padded with -1.0) or the fixed canvas width."""
        batch_w = max(p.canvas_width for p in plans) if self.dynamic_width else int(self._cfg.data.img_size[1])
        return imaging.build_crop_batch(dataset.page, plans, out_h=int(self._cfg.data.img_size[0]), batch_w=batch_w)

    # ------------------------------------------------------------------ inference + decode
    MAX_LINES_PER_FORWARD = 2048  # bounds the logits workspace of one grouped forward (101 x num_tokens floats per line)
    # ... and its activation workspace: a forward keeps ~13 fp32 copies of its [token rows] x [embed dim] stream (the q|k|v and
    # MLP hidden buffers included), so the token rows of a forward are bounded such that 13 x 4 x D x rows stays below this many
    # bytes - 1.7 M rows for the 192-wide --lite recogniser (2048 lines of 800 px: the line bound is the tighter one), 430 k for
    # parseq-large-v4_1 (1075 lines of its fixed 800 px canvas).  The same two bounds size the one-off workspace reservation.
    FORWARD_WORKSPACE_BYTES = 16 << 30

    def _token_geometry(self):
        """(token rows per line at the full canvas width, token-row budget of one forward)."""
        ph, pw = (int(v) for v in self._cfg.encoder.patch_size)
        h, w = (int(v) for v in self._cfg.data.img_size)
        dim = int(self._cfg.encoder.embed_dim)
        return (h // ph) * (w // pw), max(1, self.FORWARD_WORKSPACE_BYTES // (13 * 4 * dim))

    def _job_token_rows(self, plans):
        ph, pw = (int(v) for v in self._cfg.encoder.patch_size)
        h, w = (int(v) for v in self._cfg.data.img_size)
        width = max(p.canvas_width for p in plans) if self.dynamic_width else w
        return len(plans) * (h // ph) * (width // pw)

    def _reserve_bounds(self):
        """(lines, height, width) for ymk_model_reserve: MAX_LINES_PER_FORWARD lines as wide as the token-row budget allows."""
        ph, pw = (int(v) for v in self._cfg.encoder.patch_size)
        h, w = (int(v) for v in self._cfg.data.img_size)
        _, budget = self._token_geometry()
        cols = max(1, min(w // pw, budget // (self.MAX_LINES_PER_FORWARD * (h // ph))))
        return self.MAX_LINES_PER_FORWARD, h, cols * pw

    def _run_inference(self, data: torch.Tensor, model=None):
        model = model or self.model
        logits = model(data)
        return model.token_stats(logits)  # == softmax(-1).max(-1), without materialising the softmax

    def postprocess(self, stats, points):
        ids, probs = stats
        if isinstance(ids, torch.Tensor):
            ids, probs = ids.cpu().numpy(), probs.cpu().numpy()
        pred, score = self.tokenizer.decode_stats(ids, probs)
        pred = [unicodedata.normalize("NFKC", x) for x in pred]
        if len(points) == 0:
            return pred, score, []
        q = np.asarray(points).reshape(len(points), 4, 2)  # np.array(point) of integer quads: int64, like the reference's
        w = np.sqrt(((q[:, 0] - q[:, 1]) ** 2).sum(1).astype(np.float64))
        h = np.sqrt(((q[:, 1] - q[:, 2]) ** 2).sum(1).astype(np.float64))
        directions = np.where(h > w * 2, "vertical", "horizontal").tolist()
        return pred, score, directions

    def _forward_chunks(self, jobs):
        """jobs -> [(first job, one past the last)]: consecutive mini-batches that share a forward - at most
        MAX_LINES_PER_FORWARD lines and at most the token-row budget of `_token_geometry` (a single mini-batch over either
        bound still runs, alone; the library cuts any GEMM whose operand view would pass 4 GiB into row chunks itself)."""
        _, budget = self._token_geometry()
        lines = [len(plans) for _, plans in jobs]
        rows = [self._job_token_rows(plans) for _, plans in jobs]
        n_fwd = max(1, -(-sum(lines) // self.MAX_LINES_PER_FORWARD), -(-sum(rows) // budget))
        # forwards of about equal size: 1100 lines run as 550 + 550, not 1024 + 76
        line_target = min(self.MAX_LINES_PER_FORWARD, -(-sum(lines) // n_fwd))
        row_target = min(budget, -(-sum(rows) // n_fwd))
        chunks, start = [], 0
        while start < len(jobs):
            stop, nl, nr = start, 0, 0
            while stop < len(jobs) and (stop == start or (nl < line_target and nr < row_target
                                                          and nl + lines[stop] <= self.MAX_LINES_PER_FORWARD and nr + rows[stop] <= budget)):
                nl += lines[stop]
                nr += rows[stop]
                stop += 1
            chunks.append((start, stop))
            start = stop
        return chunks

    def _collate_jobs(self, jobs, flip=False, fixed_width=False):
        """The crop kernels of every mini-batch: one padded B x 3 x 32 x W device tensor per job."""
        tensors = []
        for dataset, plans in jobs:
            if fixed_width:
                h, w = (int(v) for v in self._cfg.data.img_size)
                tensors.append(imaging.build_crop_batch(dataset.page, plans, out_h=h, batch_w=w, flip=flip))
            else:
                tensors.append(self._collate(dataset, plans))
        return tensors

    def _forward_jobs(self, tensors, chunks, model=None):
        """One PARSeq forward per chunk (nets.PARSeq.forward_groups): every mini-batch keeps its own padded width and its
        own early-stop step count, the launches are shared.  Returns per job (ids, probs) as numpy B x S."""
        model = model or self.model
        out = [None] * len(tensors)
        for start, stop in chunks:
            part = tensors[start:stop]
            logits, out_lens, _ = model.forward_groups(part)
            ids, probs = imaging.to_host(*model.token_stats(logits))  # two DMAs into pinned memory, one wait
            row = 0
            for k, (t, n) in enumerate(zip(part, out_lens)):
                b = int(t.shape[0])
                out[start + k] = (ids[row : row + b, :n], probs[row : row + b, :n])
                row += b
        return out

    def _infer_groups(self, jobs, flip=False, fixed_width=False):
        """jobs: (dataset, plans) mini-batches - of one page or of many - through ONE PARSeq forward per
        MAX_LINES_PER_FORWARD lines.  Returns per job (ids, probs) as numpy B x S."""
        return self._forward_jobs(self._collate_jobs(jobs, flip, fixed_width), self._forward_chunks(jobs)) if jobs else []

    def _run_batch_inference(self, dataset, batches, points):
        preds, scores, directions = [], [], []
        offset = 0
        for plans, stats in zip(batches, self._infer_groups([(dataset, plans) for plans in batches])):
            pred, score, direction = self.postprocess(stats, points[offset : offset + len(plans)])
            preds.extend(pred)
            scores.extend(score)
            directions.extend(direction)
            offset += len(plans)
        return preds, scores, directions

    # ------------------------------------------------------------------ 180-degree retry (text_recognizer.py:319-350)
    def _prepare_fallback_batch(self, dataset, indices):
        """The retried crops turned by 180 degrees on the fixed-width canvas (resize_with_padding), in chunks of
        cfg.data.batch_size - the flip happens inside the crop kernel, on the same warped pixels."""
        plans = [dataset.plans[i] for i in indices]
        size = int(self._cfg.data.batch_size)
        return [plans[i : i + size] for i in range(0, len(plans), size)]

    def _apply_orientation_fallback(self, dataset, points, preds, scores, directions):
        retry = [i for i, s in enumerate(scores) if s < self.rec_orientation_fallback_thresh]
        if not retry:
            return
        retry_points = [points[i] for i in retry]
        r_preds, r_scores, r_dirs = [], [], []
        offset = 0
        batches = self._prepare_fallback_batch(dataset, retry)
        for plans, stats in zip(batches, self._infer_groups([(dataset, plans) for plans in batches], flip=True, fixed_width=True)):
            pred, score, direction = self.postprocess(stats, retry_points[offset : offset + len(plans)])
            r_preds.extend(pred)
            r_scores.extend(score)
            r_dirs.extend(direction)
            offset += len(plans)
        for j, idx in enumerate(retry):
            if r_scores[j] > scores[idx] and r_scores[j] >= self.rec_orientation_fallback_thresh:
                preds[idx], scores[idx], directions[idx] = r_preds[j], r_scores[j], r_dirs[j]

    @staticmethod
    def _with_placeholders(dataset, points, preds, scores, directions):
        """Results of the planned crops -> results aligned with `points` up to the quads the reference itself drops: a
        degenerate quad (CropSet.slots) gets ("", 0.0, its direction)."""
        if not dataset.n_degenerate:
            return preds, scores, directions
        out_p, out_s, out_d, k = [], [], [], 0
        for slot, quad in zip(dataset.slots, points):
            if slot == "ok":
                out_p.append(preds[k]), out_s.append(scores[k]), out_d.append(directions[k])
                k += 1
            elif slot == "degenerate":
                q = np.asarray(quad).reshape(4, 2)
                w, h = np.linalg.norm(q[0] - q[1]), np.linalg.norm(q[1] - q[2])
                out_p.append(""), out_s.append(0.0), out_d.append("vertical" if h > w * 2 else "horizontal")
        return out_p, out_s, out_d

    # ---- `__call__` for several pages at once, in three steps so that a pipeline can run them on different threads:
    # plan_pages (host geometry + crop kernels), forward_plan (PARSeq, owns the model), finish_plan (host decode)
    def plan_pages(self, imgs, points_list):
        """The mini-batches of every page are formed per page exactly as in `__call__` (bucketing, width budget, padding)
        and their crop tensors built; a page's result never depends on its neighbours (mini-batches never mix pages)."""
        # multi-page serving: size the PARSeq workspace once per live handle for the largest grouped forward `_forward_chunks`
        # forms (line bound and token-row bound), so that no wave - however its lines fall into mini-batches - reaches
        # hipMalloc / hipFree (~18 GB for the --lite recogniser, ~20 GB for parseq-large-v4_1, of 288 GB)
        self.model.reserve_once(*self._reserve_bounds(), self.device)
        preps = [self.preprocess(img, pts) for img, pts in zip(imgs, points_list)]
        jobs, spans = [], []
        for batches, _, dataset, _ in preps:
            spans.append((len(jobs), len(jobs) + len(batches)))
            jobs.extend((dataset, plans) for plans in batches)
        return {"preps": preps, "jobs": jobs, "spans": spans, "chunks": self._forward_chunks(jobs), "tensors": self._collate_jobs(jobs),
                "stats": None}

    def forward_plan(self, plan, model=None):
        """ALL mini-batches of all pages through shared PARSeq forwards; fills plan["stats"].  `model`: another handle with
        the same weights (`replica_model`), for callers that keep two forwards in flight."""
        plan["stats"] = self._forward_jobs(plan["tensors"], plan["chunks"], model) if plan["jobs"] else []
        plan["tensors"] = None
        return plan

    def replica_model(self, lane: int):
        """Lane 0: the module's own net.  Lane k > 0: a further PARSeq handle holding the same weights (built on first use,
        rebuilt when the module's weights change; the evaluation switches of the module's net are mirrored), with its own
        workspace - so that a second forward can be in flight on another thread and HIP stream."""
        if lane == 0:
            return self.model
        with self._replica_lock:
            rep = self._replicas.get(lane)
            if rep is None or rep._source_sd is not self.model._sd:
                if rep is not None:
                    rep.close()
                rep = type(self.model)(self.model.cfg).load_state_dict(self.model._sd).to(self.device)
                rep._source_sd = self.model._sd
                rep.tokenizer = self.tokenizer
                rep.reserve_once(*self._reserve_bounds(), self.device)
                self._replicas[lane] = rep
            if rep._conv_split != self.model._conv_split:
                rep.set_conv_split(self.model._conv_split)
            for k, v in getattr(self.model, "_extra_params", {}).items():
                if getattr(rep, "_extra_params", {}).get(k) != v:
                    rep.set_param(k, v)
            return rep

    def close_replicas(self):
        """Release the extra PARSeq handles (weights + reserved workspace) that replica_model built."""
        with self._replica_lock:
            for rep in self._replicas.values():
                rep.close()
            self._replicas = {}

    def finish_plan(self, plan):
        """Token decode, un-permutation, optional 180-degree retry (which runs further forwards: call it from the thread
        that owns the model when `rec_orientation_fallback` is on).  One TextRecognizerSchema per page."""
        stats = plan["stats"]
        results = []
        for (batches, points, dataset, order), (lo, hi) in zip(plan["preps"], plan["spans"]):
            walk = [points[i] for i in order] if order is not None else points
            preds, scores, directions = [], [], []
            offset = 0
            for plans, st in zip(batches, stats[lo:hi]):
                pred, score, direction = self.postprocess(st, walk[offset : offset + len(plans)])
                preds.extend(pred)
                scores.extend(score)
                directions.extend(direction)
                offset += len(plans)
            if order is not None:
                inverse = np.argsort(order)
                preds = [preds[i] for i in inverse]
                scores = [scores[i] for i in inverse]
                directions = [directions[i] for i in inverse]
            if self.rec_orientation_fallback:
                self._apply_orientation_fallback(dataset, points, preds, scores, directions)
            preds, scores, directions = self._with_placeholders(dataset, points, preds, scores, directions)
            results.append(TextRecognizerSchema(contents=preds, scores=scores, points=points, directions=directions))
        return results

    def recognize_pages(self, imgs, points_list):
        """`__call__` for several pages at once; one TextRecognizerSchema per page."""
        return self.finish_plan(self.forward_plan(self.plan_pages(imgs, points_list)))

    def __call__(self, img, points=None, vis=None):
        batches, points, dataset, order = self.preprocess(img, points)
        if order is not None:
            sorted_points = [points[i] for i in order]
            preds, scores, directions = self._run_batch_inference(dataset, batches, sorted_points)
            inverse = np.argsort(order)
            preds = [preds[i] for i in inverse]
            scores = [scores[i] for i in inverse]
            directions = [directions[i] for i in inverse]
        else:
            preds, scores, directions = self._run_batch_inference(dataset, batches, points)
        if self.rec_orientation_fallback:
            self._apply_orientation_fallback(dataset, points, preds, scores, directions)
        preds, scores, directions = self._with_placeholders(dataset, points, preds, scores, directions)
        results = TextRecognizerSchema(contents=preds, scores=scores, points=points, directions=directions)
        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        return results, vis

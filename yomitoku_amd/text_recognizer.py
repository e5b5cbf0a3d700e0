"""TextRecognizer module (reference text_recognizer.py:35-399): same constructor, catalog names,
config keys, batching rules and `__call__(img, points, vis) -> (TextRecognizerSchema, vis)`.

The batching rules are numerics, not plumbing (padding columns are ordinary ViT
tokens), so `np.argsort` bucketing and the width-budget rule are reproduced literally; what moves
to the MI355X is the crop warp / resize / normalise / pad (one fused kernel pair per mini-batch),
the PARSeq forward, and the softmax -> (arg-max, max-prob) reduction the tokenizer needs."""

from __future__ import annotations

import unicodedata
from typing import List, Optional

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


def load_charset(charset_path):
    with open(charset_path, "r", encoding="utf-8") as f:
        return f.read()


class ParseqTokenizer:
    """postprocessor/parseq_tokenizer.py:91-126: itos = [E] + charset + [B], [P]."""

    BOS, EOS, PAD = "[B]", "[E]", "[P]"

    def __init__(self, charset: str):
        self._itos = (self.EOS,) + tuple(charset) + (self.BOS, self.PAD)
        self._stoi = {s: i for i, s in enumerate(self._itos)}
        self.eos_id, self.bos_id, self.pad_id = (self._stoi[s] for s in (self.EOS, self.BOS, self.PAD))

    def __len__(self):
        return len(self._itos)

    def decode_stats(self, ids: np.ndarray, probs: np.ndarray):
        """Greedy decode from per-position (arg-max id, max prob): cut at the first <eos>, the score is
        the product of the kept probabilities including the <eos> one (parseq_tokenizer.py:64-88,117-126)."""
        texts, scores = [], []
        for row_ids, row_p in zip(ids, probs):
            row_ids = row_ids.tolist()
            try:
                e = row_ids.index(self.eos_id)
            except ValueError:
                e = len(row_ids)
            texts.append("".join(self._itos[i] for i in row_ids[:e]))
            scores.append(float(np.asarray(row_p[: e + 1], dtype=np.float32).prod()))
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
        self.num_parallel_batches = int(num_parallel_batches)  # mini-batches in flight (see _run_batch_inference_parallel)
        self.source_downscale = bool(source_downscale)
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
    def _run_inference(self, data: torch.Tensor, model=None):
        model = model or self.model
        logits = model(data)
        return model.token_stats(logits)  # == softmax(-1).max(-1), without materialising the softmax

    def postprocess(self, stats, points):
        ids, probs = stats
        pred, score = self.tokenizer.decode_stats(ids.cpu().numpy(), probs.cpu().numpy())
        pred = [unicodedata.normalize("NFKC", x) for x in pred]
        directions = []
        for point in points:
            point = np.array(point)
            w = np.linalg.norm(point[0] - point[1])
            h = np.linalg.norm(point[1] - point[2])
            directions.append("vertical" if h > w * 2 else "horizontal")
        return pred, score, directions

    def _run_batch_inference(self, dataset, batches, points):
        if self.num_parallel_batches > 1 and len(batches) > 1:
            return self._run_batch_inference_parallel(dataset, batches, points)
        preds, scores, directions = [], [], []
        offset = 0
        for plans in batches:
            batch_points = points[offset : offset + len(plans)]
            data = self._collate(dataset, plans)
            pred, score, direction = self.postprocess(self._run_inference(data), batch_points)
            preds.extend(pred)
            scores.extend(score)
            directions.extend(direction)
            offset += len(plans)
        return preds, scores, directions

    def _lanes(self, n):
        """n (model replica, HIP stream) pairs: mini-batches are independent, but a model handle owns one workspace
        arena, so concurrent batches need their own replica (the lite recogniser is 40 MB; packed once, kept)."""
        if getattr(self, "_lane_weights", None) is not self.model.state_dict():  # first use, or new weights were loaded
            self._lane_models, self._lane_streams = [self.model], getattr(self, "_lane_streams", [])
            self._lane_weights = self.model.state_dict()
        while len(self._lane_models) < n:
            twin = type(self.model)(cfg=self._cfg)
            twin.load_state_dict(self.model.state_dict()).to(self.device)
            self._lane_models.append(twin)
        dev = torch.device(self.device)
        while len(self._lane_streams) < n:
            self._lane_streams.append(torch.cuda.Stream(device=dev))
        return list(zip(self._lane_models[:n], self._lane_streams[:n]))

    def _run_batch_inference_parallel(self, dataset, batches, points):
        """`num_parallel_batches` mini-batches in flight (text_recognizer.py:285-317 runs them on CPU worker threads;
        there the gate is dead code).  Here each lane is a thread with its own replica and HIP stream:
        the AR decode of one batch is a chain of small launches that leaves the device mostly idle.  Outputs are
        those of the serial loop - batches do not interact - in batch order."""
        import queue
        from concurrent.futures import ThreadPoolExecutor

        n = min(int(self.num_parallel_batches), len(batches))
        free = queue.Queue()
        for lane in self._lanes(n):
            free.put(lane)
        offsets = np.cumsum([0] + [len(b) for b in batches]).tolist()
        dev = torch.device(self.device)
        issued = torch.cuda.current_stream(dev)  # the page upload / pyramid were issued here

        def work(i):
            model, stream = free.get()
            try:
                stream.wait_stream(issued)
                with torch.cuda.stream(stream):
                    data = self._collate(dataset, batches[i])
                    stats = self._run_inference(data, model)
                    out = self.postprocess(stats, points[offsets[i] : offsets[i + 1]])  # .cpu() waits for this stream
                self.model.last_ar_steps = model.last_ar_steps
                return out
            finally:
                free.put((model, stream))

        if not hasattr(self, "_lane_pool") or self._lane_pool._max_workers < n:
            self._lane_pool = ThreadPoolExecutor(max_workers=n, thread_name_prefix="ymk-rec")
        preds, scores, directions = [], [], []
        for pred, score, direction in self._lane_pool.map(work, range(len(batches))):
            preds.extend(pred)
            scores.extend(score)
            directions.extend(direction)
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
        h, w = (int(v) for v in self._cfg.data.img_size)
        r_preds, r_scores, r_dirs = [], [], []
        offset = 0
        for plans in self._prepare_fallback_batch(dataset, retry):
            data = imaging.build_crop_batch(dataset.page, plans, out_h=h, batch_w=w, flip=True)
            pred, score, direction = self.postprocess(self._run_inference(data), retry_points[offset : offset + len(plans)])
            r_preds.extend(pred)
            r_scores.extend(score)
            r_dirs.extend(direction)
            offset += len(plans)
        for j, idx in enumerate(retry):
            if r_scores[j] > scores[idx] and r_scores[j] >= self.rec_orientation_fallback_thresh:
                preds[idx], scores[idx], directions[idx] = r_preds[j], r_scores[j], r_dirs[j]

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
        results = TextRecognizerSchema(contents=preds, scores=scores, points=points, directions=directions)
        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        return results, vis

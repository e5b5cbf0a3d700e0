"""TextDetector module (reference text_detector.py:26-146): same constructor, catalog names,
config keys and `__call__(img) -> (TextDetectorSchema, vis)` contract.  Pre-processing and the
DBNet forward run on the MI355X; the DB box extraction runs in the C++ host code of libymk_hip.so."""

from __future__ import annotations

import ctypes
import threading

import numpy as np
import torch

from . import _lib, imaging
from .base import BaseModelCatalog, BaseModule
from .configs import TextDetectorDBNetConfig, TextDetectorDBNetV2_1Config, TextDetectorDBNetV2Config
from .nets import DBNet
from .schemas import TextDetectorSchema


class TextDetectorModelCatalog(BaseModelCatalog):
    def __init__(self):
        super().__init__()
        self.register("dbnet", TextDetectorDBNetConfig, DBNet)
        self.register("dbnetv2", TextDetectorDBNetV2Config, DBNet)
        self.register("dbnetv2_1", TextDetectorDBNetV2_1Config, DBNet)


class DBnetPostProcessor:
    """postprocessor/dbnet_postporcessor.py:8-27 with the box extraction delegated to ymk_db_postprocess."""

    def __init__(self, min_size, thresh, box_thresh, max_candidates, unclip_ratio):
        self.min_size = min_size
        self.thresh = thresh
        self.box_thresh = box_thresh
        self.max_candidates = max_candidates
        self.unclip_ratio = unclip_ratio
        self._pinned = {}

    def _pinned_floats(self, key, count):
        """Growable flat pinned buffer named `key`: one per (wave slot, forward of the wave) or per single-page caller -
        never one per map SHAPE, so arbitrary scan sizes do not accumulate pinned host memory."""
        host = self._pinned.get(key)
        if host is None or host.numel() < count:
            host = self._pinned[key] = torch.empty(max(int(count), 1 << 20), dtype=torch.float32, pin_memory=True)
        return host[:count]

    def maps_to_host(self, maps: torch.Tensor, slot=0) -> np.ndarray:
        """n x 1 x H x W device maps -> one pinned host array n x H x W (one DMA for the whole batch).  `slot` names the
        pinned buffer: forwards whose maps must stay valid together (the forwards of one wave, the waves in flight of
        DocumentAnalyzer.serve) use different slots."""
        shape = (maps.shape[0], maps.shape[2], maps.shape[3])
        host = self._pinned_floats(("maps", slot), shape[0] * shape[1] * shape[2]).view(shape)
        host.copy_(maps.detach().to(torch.float32).reshape(shape), non_blocking=True)
        torch.cuda.current_stream(maps.device).synchronize()
        return host.numpy()

    def __call__(self, preds, image_size):
        pred = preds["binary"] if isinstance(preds, dict) else preds
        if pred.ndim == 4:
            pred = pred[0, 0]
        if isinstance(pred, torch.Tensor):
            if pred.is_cuda:
                # one DMA into a pinned buffer (a pageable destination is copied in ~32 KB staging chunks: ~250 copy
                # kernels and 2.7 ms for the 7.6 MB map)
                host = self._pinned_floats(("page", threading.get_ident()), pred.numel()).view(pred.shape)
                host.copy_(pred.detach().to(torch.float32), non_blocking=True)
                torch.cuda.current_stream(pred.device).synchronize()
                pred = host.numpy()
            else:
                pred = pred.detach().to("cpu", torch.float32).contiguous().numpy()
        pred = np.ascontiguousarray(pred, dtype=np.float32)
        h, w = pred.shape
        height, width = image_size
        cap = int(self.max_candidates)
        quads = np.empty((cap, 4, 2), dtype=np.int16)
        scores = np.empty(cap, dtype=np.float64)
        n = ctypes.c_int()
        _lib.check(
            _lib.load().ymk_db_postprocess(pred.ctypes.data, h, w, float(self.thresh), float(self.box_thresh),
                                           int(self.min_size), cap, float(self.unclip_ratio), int(width), int(height),
                                           quads.ctypes.data, scores.ctypes.data, cap, ctypes.byref(n)),
            "ymk_db_postprocess",
        )
        return quads[: n.value].tolist(), scores[: n.value].tolist()


class TextDetector(BaseModule):
    model_catalog = TextDetectorModelCatalog()

    def __init__(self, model_name="dbnetv2_1", path_cfg=None, device="cuda", visualize=False, from_pretrained=True,
                 infer_onnx=False):
        super().__init__()
        if infer_onnx:
            raise NotImplementedError("the ONNX backend is out of scope of the MI355X path (infer_onnx=False only)")
        self.load_model(model_name, path_cfg, from_pretrained=from_pretrained)
        self.device = device
        self.visualize = visualize
        self.model.eval()
        self.post_processor = DBnetPostProcessor(**self._cfg.post_process)
        self.infer_onnx = False
        self.model.to(self.device)

    def preprocess(self, img):
        """BGR uint8 page -> fp32 1 x 3 x H' x W' on the device (one H2D copy of the uint8 page)."""
        page = img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device)
        return imaging.detector_tensor(page, self._cfg.data.shortest_size, self._cfg.data.limit_size)

    def postprocess(self, preds, image_size):
        return self.post_processor(preds, image_size)

    MAX_PAGES_PER_FORWARD = 8  # bounds the activation workspace (a 1600 x 1184 page holds ~2 GB of fp32 maps)

    def forward_pages(self, imgs, ring=0):
        """Pre-processing + DBNet forward for several pages: pages whose network input has the same size share forwards
        of up to MAX_PAGES_PER_FORWARD images (images of a batch are independent); the probability maps of a forward
        come back in one DMA.  Returns one host map (H' x W' float32, a view of a pinned buffer that the next
        forward_pages call of this module WITH THE SAME `ring` reuses) per page, in input order."""
        pages = [img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device) for img in imgs]
        cfg = self._cfg.data
        # multi-page serving: the workspace is sized once per live handle for the largest forward (MAX_PAGES_PER_FORWARD images
        # of limit_size x shortest_size, either orientation: ~16 GB of 288), so no mix of page sizes reaches hipMalloc later;
        # if that much cannot be had the model grows its workspace on demand (nets.reserve_once)
        long_side = max(32, int(cfg.limit_size) // 32 * 32)
        short_side = max(32, min(int(cfg.shortest_size), int(cfg.limit_size)) // 32 * 32)
        self.model.reserve_once(self.MAX_PAGES_PER_FORWARD, long_side, short_side, self.device)
        by_size = {}
        for i, page in enumerate(pages):
            dims = imaging.resize_shortest_edge_dims(page.shape[0], page.shape[1], cfg.shortest_size, cfg.limit_size)
            by_size.setdefault(dims, []).append(i)
        maps = [None] * len(pages)
        slot = 0
        for (oh, ow), members in by_size.items():
            per = -(-len(members) // -(-len(members) // self.MAX_PAGES_PER_FORWARD))  # forwards of equal size: 12 pages run as 6 + 6
            for start in range(0, len(members), per):
                idx = members[start : start + per]
                x = torch.empty((len(idx), 3, oh, ow), dtype=torch.float32, device=pages[idx[0]].device)
                for k, i in enumerate(idx):
                    imaging.detector_tensor(pages[i], cfg.shortest_size, cfg.limit_size, out=x[k])
                host = self.post_processor.maps_to_host(self.model(x)["binary"], (ring, slot))
                slot += 1
                for k, i in enumerate(idx):
                    maps[i] = host[k]
        return maps

    def extract_boxes(self, maps, sizes):
        """DB box extraction (C++, GIL released) for several maps concurrently; sizes: original (height, width) per
        page.  Returns one TextDetectorSchema per map."""
        from concurrent.futures import ThreadPoolExecutor

        if not hasattr(self, "_post_pool"):
            # `post_threads`: size of the pool (default 4; a sharded job sets it from its rank's core slice, distributed.thread_budget)
            self._post_pool = ThreadPoolExecutor(max_workers=max(1, int(getattr(self, "post_threads", 4))), thread_name_prefix="ymk-dbpost")
        return [TextDetectorSchema(points=quads, scores=scores)
                for quads, scores in self._post_pool.map(self.post_processor, list(maps), list(sizes))]

    def detect_pages(self, imgs):
        """`__call__` for several pages (forward_pages + extract_boxes); one TextDetectorSchema per page."""
        pages = [img if isinstance(img, torch.Tensor) else imaging.page_to_device(img, self.device) for img in imgs]
        maps = self.forward_pages(pages)
        return self.extract_boxes(maps, [tuple(int(v) for v in p.shape[:2]) for p in pages])

    def __call__(self, img):
        ori_h, ori_w = img.shape[:2]
        tensor = self.preprocess(img)
        preds = self.model(tensor)
        quads, scores = self.postprocess(preds, (ori_h, ori_w))
        results = TextDetectorSchema(points=quads, scores=scores)
        if self.visualize:
            raise NotImplementedError("visualisation is out of scope of the MI355X path (visualize=False only)")
        return results, None

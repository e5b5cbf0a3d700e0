"""Network objects behind the `self.model(tensor)` seam of the reference modules.

Each class keeps the call contract of the reference nn.Module it replaces but owns no PyTorch
parameters: the weights live inside libymk_hip.so (packed for the gfx950 kernels) and a forward
is one C-ABI call on the current HIP stream.  PyTorch is used for device memory and streams only.

  DBNet      models/dbnet_plus.py:233-246   fp32 N x 3 x H x W  ->  {"binary": N x 1 x H x W}
"""

from __future__ import annotations

import os
from collections import OrderedDict
from typing import Mapping

import torch

from . import _lib


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise _lib.YmkError(f"{what}: tensor must live on a HIP device (there is no CPU fallback)")


def load_safetensors_dir(repo_or_dir: str) -> "OrderedDict[str, torch.Tensor]":
    """Counterpart of PyTorchModelHubMixin.from_pretrained (base.py:84): `repo_or_dir` is a local directory holding
    model.safetensors, or a Hugging Face repo id resolved through the hub cache first (what a previous
    `yomitoku_download_model` / reference run left in ~/.cache/huggingface) and the network last."""
    from safetensors.torch import load_file

    path = os.path.join(repo_or_dir, "model.safetensors")
    if os.path.isfile(path):
        return OrderedDict(load_file(path))
    tried = [path]
    if not os.path.isabs(repo_or_dir) and repo_or_dir.count("/") == 1:
        try:
            from huggingface_hub import hf_hub_download
        except ImportError:  # pragma: no cover - huggingface_hub ships with the image
            hf_hub_download = None
        if hf_hub_download is not None:
            for offline in (True, False):
                try:
                    return OrderedDict(load_file(hf_hub_download(repo_or_dir, "model.safetensors", local_files_only=offline)))
                except Exception as exc:  # noqa: BLE001 - cache miss / no network: report both below
                    tried.append(f"hub {'cache' if offline else 'download'} of {repo_or_dir}: {type(exc).__name__}")
    raise FileNotFoundError(
        "pretrained weights not found (" + "; ".join(tried) + "): place model.safetensors in a local directory and point "
        "`hf_hub_repo` at it, or pass from_pretrained=False for seeded synthetic weights"
    )


class HipNet:
    """Common handle management for the C-ABI models."""

    kind = ""

    def __init__(self, cfg=None):
        self.cfg = cfg
        self._h = None
        self._device_index = None
        self._sd = None
        self.training = False
        self._conv_split = None

    # -- nn.Module-ish surface used by the reference modules (eval/to)
    def eval(self):
        return self

    def to(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise _lib.YmkError(f"{type(self).__name__} runs on MI355X only; got device {device}")
        idx = device.index if device.index is not None else torch.cuda.current_device()
        if self._h is not None and idx == self._device_index:
            return self
        if self._sd is None:
            raise _lib.YmkError("no weights loaded")
        self._build(idx)
        return self

    def load_state_dict(self, sd: Mapping[str, torch.Tensor], strict: bool = False):
        self._sd = OrderedDict((k, v) for k, v in sd.items())
        if self._h is not None:
            self._build(self._device_index)
        return self

    def state_dict(self):
        return self._sd

    def params(self) -> dict:
        return {}

    def _build(self, device_index: int):
        lib = _lib.load()
        self.close()
        h = lib.ymk_model_create(self.kind.encode(), int(device_index))
        if not h:
            _lib.check(1, f"ymk_model_create({self.kind})")
        try:
            for k, v in self.params().items():
                _lib.check(lib.ymk_model_set_param(h, k.encode(), float(v)), f"set_param {k}")
            # the evaluation switches go in BEFORE finalize: finalize builds the split weight copies for the precision the
            # model will run, so that no forward has to (include/ymk.h: ymk_model_finalize)
            if self._conv_split is not None:
                _lib.check(lib.ymk_model_set_param(h, b"conv_split", float(self._conv_split)), "set_param conv_split")
            for k, v in getattr(self, "_extra_params", {}).items():
                _lib.check(lib.ymk_model_set_param(h, k.encode(), float(v)), f"set_param {k}")
            for name, t in self._sd.items():
                if not torch.is_floating_point(t):
                    continue
                tt = t.detach().to(device="cpu", dtype=torch.float32).contiguous()
                _lib.check(
                    lib.ymk_model_set_tensor(h, name.encode(), tt.data_ptr(), tt.dim(), _lib.dims_array(tt.shape)),
                    f"set_tensor {name}",
                )
            _lib.check(lib.ymk_model_finalize(h), "ymk_model_finalize")
        except Exception:
            lib.ymk_model_destroy(h)
            raise
        self._h = h
        self._device_index = int(device_index)
        self._reserve_tried_for = None  # a new handle has no workspace reservation (reserve_once)
        self._reserved = []  # (n, h, w) bounds the live handle's workspace has been sized for (ensure_workspace)

    def set_conv_split(self, planes):
        """Operand precision of THIS model's convolutions / linear layers (include/ymk.h, "conv_split"): 16 = fp32 operands as
        two scaled fp16 planes, fp32 accumulation (what a model runs when nothing is set: the library's default since round 4),
        0 = exact fp32 MFMA, 2 / 3 = 2 / 3 bf16 planes (evaluation), None = follow the process-wide switch if one is set,
        else the default.  Takes effect from the next forward."""
        self._conv_split = None if planes is None else int(planes)
        if self._h is not None:
            _lib.check(_lib.load().ymk_model_set_param(self._h, b"conv_split", float(-1 if planes is None else int(planes))),
                       "set_param conv_split")
        return self

    def set_param(self, key: str, value: float):
        """A scalar parameter of the live handle (ymk_model_set_param); kept for rebuilds.  Used for the evaluation switches
        that may change between forwards ("conv_split", "conv_split_encoder")."""
        self._extra_params = dict(getattr(self, "_extra_params", {}), **{key: float(value)})
        if self._h is not None:
            _lib.check(_lib.load().ymk_model_set_param(self._h, key.encode(), float(value)), f"set_param {key}")
        return self

    def close(self):
        if self._h is not None:
            _lib.load().ymk_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reserve(self, n: int, h: int, w: int, device=None):
        """Size the workspace for the largest forward (ymk_model_reserve): forwards within the bound never reallocate.
        DBNet / RTDETRv2: n images of h x w; PARSeq: n lines of width <= w."""
        if self._h is None:
            self.to(device if device is not None else "cuda")
        with torch.cuda.device(self._device_index):
            _lib.check(_lib.load().ymk_model_reserve(self._h, int(n), int(h), int(w), _lib.current_stream_ptr()), "ymk_model_reserve")
        self._reserved.append((int(n), int(h), int(w)))

    def ensure_workspace(self, n: int, h: int, w: int):
        """Called by every forward with its own shape: sizes the workspace NOW - on the caller's thread and stream, outside
        the forward - unless an earlier reservation covers the shape.  With it a forward never reaches hipMalloc / hipFree
        (the library would grow the workspace itself, inside the forward, as callers of the bare C ABI may let it:
        ymk_stat("arena_grows_in_forward")); reserve_once() of the multi-page paths covers their forwards in one go."""
        for n0, h0, w0 in self._reserved:
            if n <= n0 and ((h <= h0 and w <= w0) or (self.kind == "dbnet" and h <= w0 and w <= h0)):
                return
        if getattr(self, "_reserve_tried_for", None) == self._h and not getattr(self, "_reserve_ok", True):
            return  # a reservation failed on this handle before (reserve_once): it grows on demand
        try:
            self.reserve(n, h, w)
        except _lib.YmkError:  # the forward itself reports what is wrong with the shape (or grows on demand)
            self._reserved.append((int(n), int(h), int(w)))

    def reserve_once(self, n: int, h: int, w: int, device=None) -> bool:
        """reserve() once per LIVE handle (a rebuild by load_state_dict() / to() starts over).  A reservation that fails -
        hipMalloc of the worst-case workspace, e.g. several processes sharing one GPU or a smaller HBM - is logged and the
        model falls back to growing its workspace on demand; it is not retried for this handle (a retry would free the
        working slab again on every call)."""
        if self._h is not None and getattr(self, "_reserve_tried_for", None) == self._h:
            return bool(self._reserve_ok)
        try:
            self.reserve(n, h, w, device)
            self._reserve_ok = True
        except _lib.YmkError as exc:
            import logging

            logging.getLogger("yomitoku_amd.base").warning(
                "%s: could not reserve the worst-case workspace (%d x %d x %d): %s - falling back to grow-on-demand", type(self).__name__, n, h, w, exc)
            self._reserve_ok = False
        self._reserve_tried_for = self._h
        return bool(self._reserve_ok)

    @property
    def weight_bytes(self) -> int:
        return int(_lib.load().ymk_model_weight_bytes(self._h)) if self._h else 0

    @property
    def workspace_bytes(self) -> int:
        return int(_lib.load().ymk_model_workspace_bytes(self._h)) if self._h else 0

    @classmethod
    def from_pretrained(cls, repo_or_dir: str, cfg=None):
        net = cls(cfg=cfg)
        net.load_state_dict(load_safetensors_dir(repo_or_dir))
        return net


def _cfg_get(cfg, path, default):
    cur = cfg
    try:
        for part in path.split("."):
            cur = cur[part] if isinstance(cur, dict) else getattr(cur, part)
        return cur
    except (AttributeError, KeyError, TypeError):
        return default


class PARSeq(HipNet):
    """PARSeq text recogniser (reference models/parseq.py:49-311).

    `__call__(images)` keeps the reference contract: fp32 B x 3 x 32 x W in, logits
    B x S x (num_tokens - 2) out (S = 101 when refine_iters >= 1).  `token_stats(logits)` is the
    fused replacement for `.softmax(-1)` + the arg-max/max of ParseqTokenizer.decode."""

    kind = "parseq"

    def __init__(self, cfg=None, seed: int = 1235):
        super().__init__(cfg)
        self._seed = seed
        self.tokenizer = None
        self.export_onnx = False
        self.refine_iters = int(_cfg_get(cfg, "refine_iters", 1))
        self.last_ar_steps = 0

    def params(self) -> dict:
        c = self.cfg
        patch = list(_cfg_get(c, "encoder.patch_size", [4, 8]))
        img = list(_cfg_get(c, "data.img_size", [32, 800]))
        return {
            "patch_h": patch[0], "patch_w": patch[1], "img_h": img[0], "img_w": img[1],
            "enc_dim": _cfg_get(c, "encoder.embed_dim", 192), "enc_heads": _cfg_get(c, "encoder.num_heads", 6),
            "enc_depth": _cfg_get(c, "encoder.depth", 12),
            "dec_dim": _cfg_get(c, "decoder.embed_dim", 192), "dec_heads": _cfg_get(c, "decoder.num_heads", 6),
            "dec_depth": _cfg_get(c, "decoder.depth", 1),
            "num_tokens": _cfg_get(c, "num_tokens", 7121), "max_label_length": _cfg_get(c, "max_label_length", 100),
            "refine_iters": self.refine_iters, "decode_ar": _cfg_get(c, "decode_ar", 1),
            "repetition_stop": int(bool(_cfg_get(c, "repetition_stop", True))),
            "rep_period_max": _cfg_get(c, "rep_period_max", 8), "rep_min_run_p1": _cfg_get(c, "rep_min_run_p1", 8),
            "rep_min_repeats": _cfg_get(c, "rep_min_repeats", 3),
        }

    def init_synthetic(self, seed: int | None = None, **kw):
        from .utils.synth import parseq_state_dict

        p = self.params()
        self.load_state_dict(
            parseq_state_dict(
                self._seed if seed is None else seed, patch=(p["patch_h"], p["patch_w"]), enc_dim=p["enc_dim"],
                enc_depth=p["enc_depth"], enc_mlp=int(_cfg_get(self.cfg, "encoder.mlp_ratio", 4)), dec_dim=p["dec_dim"],
                dec_mlp=int(_cfg_get(self.cfg, "decoder.mlp_ratio", 4)), num_tokens=p["num_tokens"],
                max_label_length=p["max_label_length"], img_size=(p["img_h"], p["img_w"]), **kw,
            )
        )
        return self

    def __call__(self, images: torch.Tensor):
        import ctypes

        _require_cuda(images, "PARSeq")
        if self._h is None:
            self.to(images.device)
        x = images.to(torch.float32).contiguous()
        b, c, h, w = x.shape
        lib = _lib.load()
        self.ensure_workspace(b, h, w)
        ns, nc = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.ymk_parseq_dims(self._h, ctypes.byref(ns), ctypes.byref(nc)), "ymk_parseq_dims")
        logits = torch.empty((b, ns.value, nc.value), dtype=torch.float32, device=x.device)
        out_len, ar = ctypes.c_int(), ctypes.c_int()
        with torch.cuda.device(x.device):
            _lib.check(
                lib.ymk_parseq_forward(self._h, x.data_ptr(), b, w, logits.data_ptr(), ctypes.byref(out_len),
                                       ctypes.byref(ar), _lib.current_stream_ptr()),
                "ymk_parseq_forward",
            )
        self.last_ar_steps = ar.value
        return logits[:, : out_len.value]

    def forward_groups(self, batches):
        """Several mini-batches in ONE forward (ymk_parseq_forward_groups): `batches` is a list of fp32 B_g x 3 x 32 x W_g
        device tensors, each padded to its own width.  Returns (logits [sum B_g] x S_max x C, out_lens, ar_steps) with
        per-group row counts / greedy step counts; group g's rows are logits[off_g : off_g + B_g, : out_lens[g]]."""
        import ctypes

        if not batches:
            raise ValueError("forward_groups wants at least one mini-batch")
        for t in batches:
            _require_cuda(t, "PARSeq")
        if self._h is None:
            self.to(batches[0].device)
        xs = [t.to(torch.float32).contiguous() for t in batches]
        n = len(xs)
        lib = _lib.load()
        self.ensure_workspace(sum(int(t.shape[0]) for t in xs), int(xs[0].shape[2]), max(int(t.shape[3]) for t in xs))
        ns, nc = ctypes.c_int(), ctypes.c_int()
        _lib.check(lib.ymk_parseq_dims(self._h, ctypes.byref(ns), ctypes.byref(nc)), "ymk_parseq_dims")
        total = sum(int(t.shape[0]) for t in xs)
        logits = torch.empty((total, ns.value, nc.value), dtype=torch.float32, device=xs[0].device)
        ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in xs])
        bs = (ctypes.c_int * n)(*[int(t.shape[0]) for t in xs])
        ws = (ctypes.c_int * n)(*[int(t.shape[3]) for t in xs])
        out_len, ar = (ctypes.c_int * n)(), (ctypes.c_int * n)()
        with torch.cuda.device(xs[0].device):
            _lib.check(
                lib.ymk_parseq_forward_groups(self._h, ptrs, bs, ws, n, logits.data_ptr(), out_len, ar, _lib.current_stream_ptr()),
                "ymk_parseq_forward_groups",
            )
        self.last_ar_steps = max(ar)
        return logits, list(out_len), list(ar)

    @staticmethod
    def token_stats(logits: torch.Tensor):
        """(ids int32 B x S, probs fp32 B x S): per position arg-max class and max softmax probability."""
        lib = _lib.load()
        lg = logits.contiguous()
        b, s, c = lg.shape
        ids = torch.empty((b, s), dtype=torch.int32, device=lg.device)
        probs = torch.empty((b, s), dtype=torch.float32, device=lg.device)
        with torch.cuda.device(lg.device):
            _lib.check(
                lib.ymk_parseq_token_stats(lg.data_ptr(), b * s, c, ids.data_ptr(), probs.data_ptr(),
                                           _lib.current_stream_ptr()),
                "ymk_parseq_token_stats",
            )
        return ids, probs


class RTDETRv2(HipNet):
    """RT-DETRv2 (reference models/rtdetr.py:9-22): fp32 N x 3 x 640 x 640 ->
    {"pred_logits": N x 300 x nc, "pred_boxes": N x 300 x 4}."""

    kind = "rtdetr"

    def __init__(self, cfg=None, seed: int = 1240):
        super().__init__(cfg)
        self._seed = seed

    def params(self) -> dict:
        c = self.cfg
        return {
            "num_classes": _cfg_get(c, "RTDETRTransformerv2.num_classes", 6),
            "num_queries": _cfg_get(c, "RTDETRTransformerv2.num_queries", 300),
            "num_layers": _cfg_get(c, "RTDETRTransformerv2.num_layers", 6),
            "hidden_dim": _cfg_get(c, "RTDETRTransformerv2.hidden_dim", 256),
        }

    def init_synthetic(self, seed: int | None = None):
        from .utils.synth_rtdetr import rtdetr_state_dict

        p = self.params()
        self.load_state_dict(rtdetr_state_dict(self._seed if seed is None else seed, num_classes=int(p["num_classes"])))
        return self

    def load_state_dict(self, sd, strict: bool = False):
        from .utils.synth_rtdetr import generate_anchors, sincos_pos_embed

        sd = OrderedDict(sd.items())
        size = list(_cfg_get(self.cfg, "RTDETRTransformerv2.eval_spatial_size", [640, 640]))
        if "decoder.anchors" not in sd:  # registered buffers of the reference (rtdetrv2_decoder.py:565-568)
            sd["decoder.anchors"], sd["decoder.valid_mask"] = generate_anchors(size)
        sd["decoder.valid_mask"] = sd["decoder.valid_mask"].to(torch.float32)
        # AIFI position table: the reference rebuilds it every forward (rtdetr_hybrid_encoder.py:375-378)
        sd["__aifi_pos_embed"] = sincos_pos_embed(size[1] // 32, size[0] // 32, 256)[0]
        return super().load_state_dict(sd, strict)

    def __call__(self, x: torch.Tensor, targets=None):
        _require_cuda(x, "RTDETRv2")
        if self._h is None:
            self.to(x.device)
        x = x.to(torch.float32).contiguous()
        n, c, h, w = x.shape
        self.ensure_workspace(n, h, w)
        p = self.params()
        nq, nc = int(p["num_queries"]), int(p["num_classes"])
        logits = torch.empty((n, nq, nc), dtype=torch.float32, device=x.device)
        boxes = torch.empty((n, nq, 4), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            _lib.check(
                lib.ymk_rtdetr_forward(self._h, x.data_ptr(), n, h, w, logits.data_ptr(), boxes.data_ptr(),
                                       _lib.current_stream_ptr()),
                "ymk_rtdetr_forward",
            )
        return {"pred_logits": logits, "pred_boxes": boxes}


class DBNet(HipNet):
    """DBNet++ text detector (reference models/dbnet_plus.py:233-246)."""

    kind = "dbnet"

    def __init__(self, cfg=None, seed: int = 1234):
        super().__init__(cfg)
        self._seed = seed

    def init_synthetic(self, seed: int | None = None):
        from .utils.synth import dbnet_state_dict

        hidden = 256
        if self.cfg is not None:
            try:
                hidden = int(self.cfg.decoder.hidden_dim)
            except Exception:
                pass
        self.load_state_dict(dbnet_state_dict(self._seed if seed is None else seed, hidden))
        return self

    def __call__(self, tensor: torch.Tensor):
        _require_cuda(tensor, "DBNet")
        if self._h is None:
            self.to(tensor.device)
        x = tensor.to(torch.float32).contiguous()
        n, c, h, w = x.shape
        if c != 3:
            raise _lib.YmkError("DBNet wants N x 3 x H x W")
        self.ensure_workspace(n, h, w)
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            _lib.check(
                lib.ymk_dbnet_forward(self._h, x.data_ptr(), n, h, w, out.data_ptr(), _lib.current_stream_ptr()),
                "ymk_dbnet_forward",
            )
        return OrderedDict(binary=out)

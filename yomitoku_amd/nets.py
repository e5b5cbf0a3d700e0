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
    """Offline counterpart of PyTorchModelHubMixin.from_pretrained (base.py:84): `repo_or_dir`
    must be a local directory holding model.safetensors (what `yomitoku_download_model` writes)."""
    path = os.path.join(repo_or_dir, "model.safetensors")
    if not os.path.isfile(path):
        raise FileNotFoundError(
            f"{path} not found: pretrained weights must be available locally (no network); "
            "pass from_pretrained=False for seeded synthetic weights"
        )
    from safetensors.torch import load_file

    return OrderedDict(load_file(path))


class HipNet:
    """Common handle management for the C-ABI models."""

    kind = ""

    def __init__(self, cfg=None):
        self.cfg = cfg
        self._h = None
        self._device_index = None
        self._sd = None
        self.training = False

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

    def close(self):
        if self._h is not None:
            _lib.load().ymk_model_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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
        out = torch.empty((n, 1, h, w), dtype=torch.float32, device=x.device)
        lib = _lib.load()
        with torch.cuda.device(x.device):
            _lib.check(
                lib.ymk_dbnet_forward(self._h, x.data_ptr(), n, h, w, out.data_ptr(), _lib.current_stream_ptr()),
                "ymk_dbnet_forward",
            )
        return OrderedDict(binary=out)

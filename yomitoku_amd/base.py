"""Module base classes: catalog of model variants, config loading, the timing/exception observer
and device handling - the reference's `base.py:36-142` surface, on top of `yomitoku_amd.config`."""

from __future__ import annotations

import logging
import time
import warnings

import torch
from pydantic import BaseModel, ConfigDict

from . import config as _config
from .config import load_config, load_yaml_config  # noqa: F401  (re-exported like the reference)


def set_logger(name, level="INFO"):
    # utils/logger.py:5-15 (also silences warnings, as the reference does)
    logger = logging.getLogger(name)
    logger.setLevel(level)
    if not logger.handlers:
        handler = logging.StreamHandler()
        handler.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
        logger.addHandler(handler)
    warnings.filterwarnings("ignore")
    return logger


logger = set_logger(__name__, "INFO")


def observer(cls, func):
    """Wall-clock timing of `__call__` plus log-and-reraise of exceptions (base.py:36-48)."""

    def wrapper(*args, **kwargs):
        try:
            start = time.time()
            result = func(*args, **kwargs)
            logger.info(f"{cls.__name__} {func.__name__} elapsed_time: {time.time() - start}")
        except Exception as e:
            logger.error(f"Error occurred in {cls.__name__} {func.__name__}: {e}")
            raise e
        return result

    wrapper.__wrapped_by_observer__ = True
    return wrapper


class BaseSchema(BaseModel):
    model_config = ConfigDict(extra="forbid", validate_assignment=True)

    def dict(self, *a, **k):  # pydantic-v1 spelling used throughout the reference
        return self.model_dump(*a, **k)

    def to_json(self, out_path: str, **kwargs):
        import json
        import os

        d = os.path.dirname(out_path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(out_path, "w", encoding="utf-8") as f:
            json.dump(self.model_dump(), f, ensure_ascii=False, indent=kwargs.get("indent", 4))


class BaseModelCatalog:
    def __init__(self):
        self.catalog = {}

    def get(self, model_name):
        model_name = model_name.lower()
        if model_name in self.catalog:
            return self.catalog[model_name]
        raise ValueError(f"Unknown model: {model_name}")

    def register(self, model_name, config, model):
        if model_name in self.catalog:
            raise ValueError(f"{model_name} is already registered.")
        self.catalog[model_name] = (config, model)

    def list_model(self):
        return list(self.catalog.keys())


class BaseModule:
    model_catalog = None

    def __init__(self):
        if self.model_catalog is None:
            raise NotImplementedError
        if not issubclass(self.model_catalog.__class__, BaseModelCatalog):
            raise ValueError(f"{self.model_catalog.__class__} is not SubClass BaseModelCatalog.")
        if len(self.model_catalog.list_model()) == 0:
            raise ValueError("No model is registered.")

    def __new__(cls, *args, **kwds):
        logger.info(f"Initialize {cls.__name__}")
        if not getattr(cls.__call__, "__wrapped_by_observer__", False):  # wrap once per class
            cls.__call__ = observer(cls, cls.__call__)
        return super().__new__(cls)

    def load_model(self, name, path_cfg, from_pretrained=True):
        """base.py:80-86.  `from_pretrained=True` needs `hf_hub_repo` to be a local directory with
        model.safetensors (no network here); otherwise a seeded synthetic checkpoint is drawn."""
        default_cfg, Net = self.model_catalog.get(name)
        self._cfg = load_config(default_cfg, path_cfg)
        if from_pretrained:
            self.model = Net.from_pretrained(self._cfg.hf_hub_repo, cfg=self._cfg)
        else:
            self.model = Net(cfg=self._cfg).init_synthetic()

    def save_config(self, path_cfg):
        _config.save(self._cfg, path_cfg)

    def log_config(self):
        logger.info(_config.to_yaml(self._cfg))

    @classmethod
    def catalog(cls):
        logger.info(f"{cls.__name__} Implemented Models")
        logger.info(" ".join(cls.model_catalog.list_model()) + " ")

    @property
    def device(self):
        return self._device

    @device.setter
    def device(self, device):
        """The HIP path has no CPU fallback: unlike the reference (base.py:107-121, cuda -> cpu with a
        warning) a missing GPU is an error the moment a network is moved to the device."""
        device = str(device)
        if "cuda" in device:
            self._device = torch.device(device)
            if not torch.cuda.is_available():
                logger.warning("No HIP device is visible; yomitoku_amd has no CPU fallback and will fail at first use.")
        else:
            raise ValueError(f"yomitoku_amd runs on MI355X (device='cuda[:N]') only, got {device!r}")

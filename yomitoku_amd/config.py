"""Structured configuration without OmegaConf.

The reference builds every module's config with `OmegaConf.structured(dataclass)` merged with an
optional YAML file (base.py:15-33).  omegaconf is not a dependency here; `ConfigNode` provides the
subset the path relies on:

  * attribute and item access, nested nodes, lists as plain Python lists;
  * `**cfg.section` expansion (mapping protocol);
  * `getattr(cfg, "missing", default)` returns the default - a missing key raises an AttributeError
    subclass (parseq.py:93-96, text_recognizer.py:167-169, layout_parser.py:99);
  * merging a YAML over a structured default rejects keys the dataclass does not declare;
  * save / dump to YAML.
"""

from __future__ import annotations

import dataclasses
from collections.abc import Mapping
from pathlib import Path
from typing import Any, Union

import yaml


class ConfigKeyError(AttributeError, KeyError):
    """Raised for a key the structured config does not declare."""


class ConfigNode(Mapping):
    __slots__ = ("_data", "_struct")

    def __init__(self, data: dict | None = None, struct: bool = True):
        object.__setattr__(self, "_data", {})
        object.__setattr__(self, "_struct", struct)
        for k, v in (data or {}).items():
            self._data[k] = _wrap(v, struct)

    # -- mapping protocol (enables ** expansion)
    def __getitem__(self, key):
        try:
            return self._data[key]
        except KeyError:
            raise ConfigKeyError(f"Missing key {key}") from None

    def __iter__(self):
        return iter(self._data)

    def __len__(self):
        return len(self._data)

    def __contains__(self, key):
        return key in self._data

    # -- attribute protocol
    def __getattr__(self, key):
        if key.startswith("__") and key.endswith("__"):
            raise AttributeError(key)
        try:
            return self._data[key]
        except KeyError:
            raise ConfigKeyError(f"Missing key {key}") from None

    def __setattr__(self, key, value):
        self[key] = value

    def __setitem__(self, key, value):
        if self._struct and key not in self._data:
            raise ConfigKeyError(f"Key '{key}' is not in struct")
        self._data[key] = _wrap(value, self._struct)

    def __repr__(self):
        return f"ConfigNode({to_container(self)!r})"

    def __deepcopy__(self, memo):
        return ConfigNode(to_container(self), self._struct)

    def __reduce__(self):
        return (ConfigNode, (to_container(self), self._struct))


def _wrap(v, struct):
    if isinstance(v, ConfigNode):
        return ConfigNode(to_container(v), struct)
    if dataclasses.is_dataclass(v) and not isinstance(v, type):
        return ConfigNode(_from_dataclass(v), struct)
    if isinstance(v, Mapping):
        return ConfigNode(dict(v), struct)
    if isinstance(v, (list, tuple)):
        return [_wrap(x, struct) for x in v]
    return v


def _from_dataclass(obj) -> dict:
    if isinstance(obj, type):
        obj = obj()
    out = {}
    for f in dataclasses.fields(obj):
        v = getattr(obj, f.name)
        if dataclasses.is_dataclass(v):
            v = _from_dataclass(v)
        elif isinstance(v, (list, tuple)):
            v = [(_from_dataclass(x) if dataclasses.is_dataclass(x) else x) for x in v]
        out[f.name] = v
    return out


def to_container(node) -> Any:
    """Plain dict / list / scalar copy of a config tree."""
    if isinstance(node, ConfigNode):
        return {k: to_container(v) for k, v in node._data.items()}
    if isinstance(node, Mapping):
        return {k: to_container(v) for k, v in node.items()}
    if isinstance(node, (list, tuple)):
        return [to_container(v) for v in node]
    return node


def structured(default) -> ConfigNode:
    """Config tree from a dataclass type or instance (struct mode: unknown keys are errors)."""
    return ConfigNode(_from_dataclass(default), struct=True)


def _coerce(old, new, path):
    """Keep the declared scalar type where the YAML value is compatible (OmegaConf validates types)."""
    if old is None or new is None or isinstance(old, (ConfigNode, list)):
        return new
    if isinstance(old, bool):
        if isinstance(new, bool):
            return new
        raise ValueError(f"{path}: expected bool, got {new!r}")
    if isinstance(old, int) and not isinstance(old, bool):
        if isinstance(new, bool):
            raise ValueError(f"{path}: expected int, got {new!r}")
        if isinstance(new, int):
            return new
        if isinstance(new, float) and float(new).is_integer():
            return int(new)
        if isinstance(new, str):
            return int(new)
        raise ValueError(f"{path}: expected int, got {new!r}")
    if isinstance(old, float):
        if isinstance(new, (int, float)) and not isinstance(new, bool):
            return float(new)
        if isinstance(new, str):
            return float(new)
        raise ValueError(f"{path}: expected float, got {new!r}")
    if isinstance(old, str):
        return str(new)
    return new


def merge(base: ConfigNode, override, _path="") -> ConfigNode:
    """Recursive merge (returns a new tree).  Lists are replaced, mappings are merged."""
    out = ConfigNode(to_container(base), base._struct)
    items = override.items() if isinstance(override, Mapping) else []
    for k, v in items:
        path = f"{_path}.{k}" if _path else str(k)
        if k not in out._data:
            if out._struct:
                raise ConfigKeyError(f"Key '{path}' not in the structured config")
            out._data[k] = _wrap(v, out._struct)
            continue
        cur = out._data[k]
        if isinstance(cur, ConfigNode) and isinstance(v, Mapping):
            out._data[k] = merge(cur, v, path)
        else:
            out._data[k] = _wrap(_coerce(cur, to_container(v), path), out._struct)
    return out


def load_yaml_config(path_config: Union[str, Path]) -> ConfigNode:
    path_config = Path(path_config)
    if not path_config.exists():
        raise FileNotFoundError(f"Config file not found: {path_config}")
    try:
        with open(path_config, "r", encoding="utf-8") as f:
            data = yaml.safe_load(f)
    except (UnicodeDecodeError, yaml.YAMLError) as e:
        raise ValueError(f"Invalid config file: {path_config}") from e
    if data is None:
        data = {}
    if not isinstance(data, dict):
        raise ValueError(f"Invalid config file (not a mapping): {path_config}")
    return ConfigNode(data, struct=False)


def load_config(default_config, path_config: Union[str, None] = None) -> ConfigNode:
    """base.py:25-33: structured default merged with an optional YAML override."""
    cfg = structured(default_config)
    if path_config is not None:
        cfg = merge(cfg, load_yaml_config(path_config))
    return cfg


def to_yaml(cfg) -> str:
    return yaml.safe_dump(to_container(cfg), sort_keys=False, allow_unicode=True)


def save(cfg, path) -> None:
    with open(path, "w", encoding="utf-8") as f:
        f.write(to_yaml(cfg))

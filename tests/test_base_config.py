"""Config / catalog / module base: the behaviours the reference's own tests/test_base.py checks, and
every default config value against the reference's dataclasses (oracle/pin_against_reference.py configs ->
tests/golden/configs.json).  No GPU: nothing here builds a network."""
import json
import os

import pytest

GOLD = os.path.join(os.path.dirname(__file__), "golden", "configs.json")


def _clean(v):
    if isinstance(v, dict):
        return {k: _clean(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_clean(x) for x in v]
    if isinstance(v, str) and os.sep in v and os.path.splitext(v)[1] in (".txt", ".ttf", ".otf"):
        return "<resource>/" + os.path.basename(v)
    return v


def test_default_configs_equal_the_reference():
    from yomitoku_amd import configs
    from yomitoku_amd.config import structured, to_container

    with open(GOLD) as f:
        gold = json.load(f)
    assert len(gold) == 12
    for name, want in gold.items():
        got = _clean(to_container(structured(getattr(configs, name))))
        assert got == want, name


def test_resources_named_by_the_configs_exist():
    from yomitoku_amd import configs
    from yomitoku_amd.config import structured

    for name in ("TextRecognizerPARSeqConfig", "TextRecognizerPARSeqTinyDynwV4Config", "TextRecognizerPARSeqLargeV41Config"):
        cfg = structured(getattr(configs, name))
        assert os.path.isfile(cfg.charset), cfg.charset
        with open(cfg.charset, encoding="utf-8") as f:
            assert len(f.read()) + 3 == cfg.num_tokens  # [E] + charset + [B], [P]


def test_load_yaml_config(tmp_path):
    from yomitoku_amd.config import load_config, load_yaml_config
    from yomitoku_amd.configs import LayoutParserRTDETRv2Config

    with pytest.raises(FileNotFoundError):
        load_yaml_config(tmp_path / "dummy.yaml")
    binary = tmp_path / "test.jpg"
    binary.write_bytes(bytes(range(256)) * 4)
    with pytest.raises(ValueError):
        load_yaml_config(binary)
    override = tmp_path / "layout_parser.yaml"
    override.write_text("thresh_score: 0.8\n")
    assert load_yaml_config(override).thresh_score == 0.8
    cfg = load_config(LayoutParserRTDETRv2Config, str(override))
    assert cfg.thresh_score == 0.8
    assert cfg.hf_hub_repo == "KotaroKinoshita/yomitoku-layout-parser-rtdtrv2-open-beta"
    nested = tmp_path / "nested.yaml"
    nested.write_text("RTDETRTransformerv2:\n  num_queries: 100\n")
    cfg = load_config(LayoutParserRTDETRv2Config, str(nested))
    assert cfg.RTDETRTransformerv2.num_queries == 100 and cfg.RTDETRTransformerv2.num_layers == 6  # siblings keep defaults


def test_catalog_and_module_contract(tmp_path):
    from yomitoku_amd import base
    from yomitoku_amd.config import load_yaml_config
    from yomitoku_amd.configs import LayoutParserRTDETRv2Config
    from yomitoku_amd.nets import RTDETRv2

    class Catalog(base.BaseModelCatalog):
        def __init__(self):
            super().__init__()
            self.register("test", LayoutParserRTDETRv2Config, RTDETRv2)

    catalog = Catalog()
    assert catalog.list_model() == ["test"]
    with pytest.raises(ValueError):
        catalog.get("dummy")
    with pytest.raises(ValueError):
        catalog.register("test", None, None)

    class Module(base.BaseModule):
        model_catalog = Catalog()
        calls = 0

        def __init__(self):
            super().__init__()

        def __call__(self):
            Module.calls += 1
            return "ran"

    module = Module()
    module.load_model("test", None, from_pretrained=False)  # no weights exist offline: seeded synthetic draw, host side only
    assert isinstance(module.model, RTDETRv2)
    module.save_config(tmp_path / "config.yaml")
    assert load_yaml_config(tmp_path / "config.yaml").hf_hub_repo == LayoutParserRTDETRv2Config().hf_hub_repo
    module.log_config()
    module.catalog()
    assert module() == "ran" and Module.calls == 1  # __call__ goes through the observer wrapper and still returns

    class Invalid(base.BaseModule):
        def __init__(self):
            super().__init__()

    with pytest.raises(NotImplementedError):
        Invalid()
    with pytest.raises(Exception):
        module.device = "cpu"  # the MI355X path has no CPU fallback


def test_pretrained_weights_resolution(tmp_path, monkeypatch):
    """from_pretrained: a local directory, then the Hugging Face hub cache (offline), then a loud error."""
    import torch
    from safetensors.torch import save_file

    from yomitoku_amd.nets import load_safetensors_dir

    local = tmp_path / "weights"
    local.mkdir()
    save_file({"a.weight": torch.arange(6, dtype=torch.float32).reshape(2, 3)}, str(local / "model.safetensors"))
    sd = load_safetensors_dir(str(local))
    assert list(sd) == ["a.weight"] and sd["a.weight"].shape == (2, 3)
    # hub cache layout: models--<org>--<name>/{refs/main, snapshots/<rev>/model.safetensors}
    hub = tmp_path / "hf" / "hub" / "models--SomeOrg--some-model"
    (hub / "refs").mkdir(parents=True)
    (hub / "refs" / "main").write_text("abc123")
    snap = hub / "snapshots" / "abc123"
    snap.mkdir(parents=True)
    save_file({"b.bias": torch.ones(4)}, str(snap / "model.safetensors"))
    monkeypatch.setenv("HF_HOME", str(tmp_path / "hf"))
    monkeypatch.setenv("HF_HUB_CACHE", str(tmp_path / "hf" / "hub"))
    monkeypatch.setenv("HF_HUB_OFFLINE", "1")
    import importlib

    import huggingface_hub.constants as hc

    importlib.reload(hc)
    try:
        import huggingface_hub.file_download as fd

        importlib.reload(fd)
        sd = load_safetensors_dir("SomeOrg/some-model")
        assert list(sd) == ["b.bias"]
        with pytest.raises(FileNotFoundError, match="pretrained weights not found"):  # offline mode: no network attempt
            load_safetensors_dir("SomeOrg/missing-model")
    finally:
        monkeypatch.undo()
        importlib.reload(hc)
        importlib.reload(fd)


def test_ensure_workspace_reserves_only_what_no_earlier_reservation_covers():
    """nets.HipNet.ensure_workspace (called by every forward with its own shape): a shape inside an earlier reservation costs
    nothing; a larger one reserves once; DBNet's reservations cover either orientation; a handle whose big reservation failed
    (reserve_once) grows on demand and is left alone.  No device: `reserve` is replaced by a recorder."""
    from yomitoku_amd import nets

    calls = []

    class Probe(nets.HipNet):
        kind = "rtdetr"

        def reserve(self, n, h, w, device=None):
            calls.append((n, h, w))
            self._reserved.append((int(n), int(h), int(w)))

    net = Probe()
    net._h, net._reserved = object(), []
    net.ensure_workspace(1, 640, 640)
    net.ensure_workspace(1, 640, 640)
    net.ensure_workspace(4, 640, 640)
    net.ensure_workspace(2, 640, 640)
    assert calls == [(1, 640, 640), (4, 640, 640)]
    det = Probe()
    det.kind, det._h, det._reserved = "dbnet", object(), []
    det.ensure_workspace(8, 1600, 1280)
    det.ensure_workspace(8, 1280, 1600)   # the other orientation of the same reservation
    det.ensure_workspace(2, 1184, 1600)
    det.ensure_workspace(8, 1600, 1600)   # wider than anything reserved: a new one
    assert calls[2:] == [(8, 1600, 1280), (8, 1600, 1600)]
    sad = Probe()
    sad._h, sad._reserved = object(), []
    sad._reserve_tried_for, sad._reserve_ok = sad._h, False
    sad.ensure_workspace(64, 640, 640)
    assert len(calls) == 4

"""Default configurations of the path's model variants.

Same key names, nesting and default values as the reference's `configs/cfg_*.py` dataclasses (the
YAML override surface users already have: `data.shortest_size`, `post_process.thresh`,
`data.width_budget`, `thresh_score`, ...), declared here through small factories instead of one
file per variant.  `resource` paths point at this package's copy of the character sets.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field, make_dataclass
from typing import List, Optional

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESOURCE_DIR = os.path.join(PKG_DIR, "resource")


def _lst(*v):
    return field(default_factory=lambda: list(v))


def _sub(cls, **over):
    """field(default_factory=...) producing `cls` with some defaults overridden."""
    return field(default_factory=lambda: cls(**over))


# ------------------------------------------------------------------ text detector (cfg_text_detector_dbnet*.py)
@dataclass
class DetBackBone:
    name: str = "resnet50"
    dilation: bool = True


@dataclass
class DetDecoder:
    in_channels: List[int] = _lst(256, 512, 1024, 2048)
    hidden_dim: int = 256
    adaptive: bool = True
    serial: bool = True
    smooth: bool = False
    k: int = 50


@dataclass
class DetData:
    shortest_size: int = 1280
    limit_size: int = 1600


@dataclass
class DetPostProcess:
    min_size: int = 2
    thresh: float = 0.3
    box_thresh: float = 0.4
    max_candidates: int = 1500
    unclip_ratio: float = 3.5


@dataclass
class DetVisualize:
    color: List[int] = _lst(0, 255, 0)
    heatmap: bool = False


def _detector(name, repo, thresh, box_thresh, unclip):
    return make_dataclass(
        name,
        [
            ("hf_hub_repo", str, repo),
            ("backbone", DetBackBone, _sub(DetBackBone)),
            ("decoder", DetDecoder, _sub(DetDecoder)),
            ("data", DetData, _sub(DetData)),
            ("post_process", DetPostProcess, _sub(DetPostProcess, thresh=thresh, box_thresh=box_thresh, unclip_ratio=unclip)),
            ("visualize", DetVisualize, _sub(DetVisualize)),
        ],
    )


TextDetectorDBNetConfig = _detector("TextDetectorDBNetConfig", "KotaroKinoshita/yomitoku-text-detector-dbnet-open-beta", 0.15, 0.5, 7.0)
TextDetectorDBNetV2Config = _detector("TextDetectorDBNetV2Config", "KotaroKinoshita/yomitoku-text-detector-dbnet-v2", 0.2, 0.5, 5.0)
TextDetectorDBNetV2_1Config = _detector("TextDetectorDBNetV2_1Config", "KotaroKinoshita/yomitoku-text-detector-dbnet-v2_1", 0.3, 0.4, 3.5)


# ------------------------------------------------------------------ text recogniser (cfg_text_recognizer_parseq*.py)
@dataclass
class RecData:
    num_workers: int = 4
    batch_size: int = 128
    img_size: List[int] = _lst(32, 800)


@dataclass
class RecDataDynw:
    num_workers: int = 4
    batch_size: int = 10
    img_size: List[int] = _lst(32, 800)
    width_budget: int = 8000
    max_batch_size: Optional[int] = 64


@dataclass
class RecEncoder:
    patch_size: List[int] = _lst(8, 8)
    num_heads: int = 8
    embed_dim: int = 512
    mlp_ratio: int = 4
    depth: int = 12


@dataclass
class RecDecoder:
    embed_dim: int = 512
    num_heads: int = 8
    mlp_ratio: int = 4
    depth: int = 1


@dataclass
class RecVisualize:
    font: str = os.path.join(RESOURCE_DIR, "MPLUS1p-Medium.ttf")
    color: List[int] = _lst(0, 0, 255)
    font_size: int = 18


def _recognizer(name, repo, charset, num_tokens, dim, heads, patch, depth=12, max_label=100, img_w=800, data_cls=RecData,
                font="MPLUS1p-Medium.ttf"):
    return make_dataclass(
        name,
        [
            ("hf_hub_repo", str, repo),
            ("charset", str, os.path.join(RESOURCE_DIR, charset)),
            ("num_tokens", int, num_tokens),
            ("max_label_length", int, max_label),
            ("decode_ar", int, 1),
            ("refine_iters", int, 1),
            ("rec_orientation_fallback", bool, False),
            ("rec_orientation_fallback_thresh", float, 0.75),
            ("data", data_cls, _sub(data_cls, img_size=[32, img_w])),
            ("encoder", RecEncoder, _sub(RecEncoder, patch_size=list(patch), num_heads=heads, embed_dim=dim, depth=depth)),
            ("decoder", RecDecoder, _sub(RecDecoder, embed_dim=dim, num_heads=heads)),
            ("visualize", RecVisualize, _sub(RecVisualize, font=os.path.join(RESOURCE_DIR, font))),
        ],
    )


_HF = "KotaroKinoshita/yomitoku-text-recognizer-"
TextRecognizerPARSeqConfig = _recognizer("TextRecognizerPARSeqConfig", _HF + "parseq-open-beta", "charset.txt", 7312, 512, 8, (8, 8))
TextRecognizerPARSeqV2Config = _recognizer("TextRecognizerPARSeqV2Config", _HF + "parseq-middle-v2", "charset.txt", 7312, 512, 8, (8, 8))
TextRecognizerPARSeqSmallConfig = _recognizer("TextRecognizerPARSeqSmallConfig", _HF + "parseq-small-open-beta", "charset.txt", 7312,
                                              384, 8, (16, 16), depth=9)
TextRecognizerPARSeqTinyConfig = _recognizer("TextRecognizerPARSeqTinyConfig", _HF + "parseq-tiny", "charsetv2.txt", 7121, 368, 8,
                                             (8, 16), max_label=50, img_w=400, font="ShipporiMinchoB1-Bold.ttf")
TextRecognizerPARSeqLargeV41Config = _recognizer("TextRecognizerPARSeqLargeV41Config", _HF + "parseq-large-v4_1", "charsetv2.txt",
                                                 7121, 768, 8, (8, 8), font="ShipporiMinchoB1-Bold.ttf")
TextRecognizerPARSeqTinyDynwV4Config = _recognizer("TextRecognizerPARSeqTinyDynwV4Config", _HF + "parseq-tiny-dynw-v4",
                                                   "charsetv2.txt", 7121, 192, 6, (4, 8), data_cls=RecDataDynw,
                                                   font="ShipporiMinchoB1-Bold.ttf")


# ------------------------------------------------------------------ RT-DETRv2 (cfg_layout_parser_rtdtrv2*.py, cfg_table_structure_*.py)
@dataclass
class RtData:
    img_size: List[int] = _lst(640, 640)


@dataclass
class RtBackBone:
    depth: int = 50
    variant: str = "d"
    freeze_at: int = 0
    return_idx: List[int] = _lst(1, 2, 3)
    num_stages: int = 4
    freeze_norm: bool = True


@dataclass
class RtEncoder:
    in_channels: List[int] = _lst(512, 1024, 2048)
    feat_strides: List[int] = _lst(8, 16, 32)
    hidden_dim: int = 256
    use_encoder_idx: List[int] = _lst(2)
    num_encoder_layers: int = 1
    nhead: int = 8
    dim_feedforward: int = 1024
    dropout: float = 0.0
    enc_act: str = "gelu"
    expansion: float = 1.0
    depth_mult: int = 1
    act: str = "silu"


@dataclass
class RtDecoder:
    num_classes: int = 6
    feat_channels: List[int] = _lst(256, 256, 256)
    feat_strides: List[int] = _lst(8, 16, 32)
    hidden_dim: int = 256
    num_levels: int = 3
    num_layers: int = 6
    num_queries: int = 300
    num_denoising: int = 100
    label_noise_ratio: float = 0.5
    box_noise_scale: float = 1.0
    eval_spatial_size: List[int] = _lst(640, 640)
    eval_idx: int = -1
    num_points: List[int] = _lst(4, 4, 4)
    cross_attn_method: str = "default"
    query_select_method: str = "default"


_LAYOUT_CATEGORY = ["tables", "figures", "paragraphs", "section_headings", "page_header", "page_footer"]
_LAYOUT_ROLE = ["section_headings", "page_header", "page_footer"]


def _layout(name, repo):
    return make_dataclass(
        name,
        [
            ("hf_hub_repo", str, repo),
            ("thresh_score", float, 0.5),
            ("data", RtData, _sub(RtData)),
            ("PResNet", RtBackBone, _sub(RtBackBone)),
            ("HybridEncoder", RtEncoder, _sub(RtEncoder)),
            ("RTDETRTransformerv2", RtDecoder, _sub(RtDecoder)),
            ("category", List[str], field(default_factory=lambda: list(_LAYOUT_CATEGORY))),
            ("role", List[str], field(default_factory=lambda: list(_LAYOUT_ROLE))),
        ],
    )


LayoutParserRTDETRv2Config = _layout("LayoutParserRTDETRv2Config", "KotaroKinoshita/yomitoku-layout-parser-rtdtrv2-open-beta")
LayoutParserRTDETRv2V2Config = _layout("LayoutParserRTDETRv2V2Config", "KotaroKinoshita/yomitoku-layout-parser-rtdtrv2-v2")

TableStructureRecognizerRTDETRv2Config = make_dataclass(
    "TableStructureRecognizerRTDETRv2Config",
    [
        ("hf_hub_repo", str, "KotaroKinoshita/yomitoku-table-structure-recognizer-rtdtrv2-open-beta"),
        ("thresh_score", float, 0.4),
        ("data", RtData, _sub(RtData)),
        ("PResNet", RtBackBone, _sub(RtBackBone)),
        ("HybridEncoder", RtEncoder, _sub(RtEncoder)),
        ("RTDETRTransformerv2", RtDecoder, _sub(RtDecoder, num_classes=3)),
        ("category", List[str], field(default_factory=lambda: ["row", "col", "span"])),
    ],
)

# configs/cfg_table_cell_parser_rtdtrv2.py: the same RT-DETRv2 at 960 x 960 with 1500 queries (dense tables reach ~1000 cells)
TableCellParserRTDETRv2Config = make_dataclass(
    "TableCellParserRTDETRv2Config",
    [
        ("hf_hub_repo", str, "KotaroKinoshita/yomitoku-cell-detector-rtdtrv2-v1"),
        ("thresh_score", float, 0.5),
        ("data", RtData, _sub(RtData, img_size=[960, 960])),
        ("PResNet", RtBackBone, _sub(RtBackBone)),
        ("HybridEncoder", RtEncoder, _sub(RtEncoder)),
        ("RTDETRTransformerv2", RtDecoder, _sub(RtDecoder, num_classes=6, num_queries=1500, eval_spatial_size=[960, 960])),
        ("category", List[str], field(default_factory=lambda: ["table", "cell", "header", "empty", "kv_item", "grid"])),
    ],
)

DEFAULT_CONFIGS = [
    TextRecognizerPARSeqLargeV41Config,
    TextDetectorDBNetV2_1Config,
    LayoutParserRTDETRv2V2Config,
    TableStructureRecognizerRTDETRv2Config,
]

__all__ = [
    "TextDetectorDBNetConfig", "TextDetectorDBNetV2Config", "TextDetectorDBNetV2_1Config",
    "TextRecognizerPARSeqConfig", "TextRecognizerPARSeqTinyConfig", "TextRecognizerPARSeqSmallConfig",
    "TextRecognizerPARSeqV2Config", "TextRecognizerPARSeqLargeV41Config", "TextRecognizerPARSeqTinyDynwV4Config",
    "LayoutParserRTDETRv2Config", "LayoutParserRTDETRv2V2Config", "TableStructureRecognizerRTDETRv2Config",
    "TableCellParserRTDETRv2Config", "DEFAULT_CONFIGS",
]

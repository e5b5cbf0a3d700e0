"""Discrete outputs of whole pages with split-operand convolutions (SPLIT env: 2 / 3 = bf16 planes, 16 = two scaled fp16
planes, the default) on the detector and the two RT-DETRv2 nets (recogniser exact unless ALL=1) against the exact-fp32 run of the same analyzer: every string / box / order / id leaf of the DocumentAnalyzerSchemas
of N synthetic pages, through DocumentAnalyzer.serve.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from split_eval import schema_diff  # noqa: E402


def main():
    from yomitoku_amd import DocumentAnalyzer
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    lite = {"ocr": {"text_detector": {"from_pretrained": False},
                    "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True, "batch_bucketing": True, "source_downscale": True}},
            "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}}}
    an = DocumentAnalyzer(configs=lite, device="cuda:0")
    # checkpoints and pages of tools/e2e_oracle_eval.py: class logits spread and biased per class (from the CPU oracle's logits
    # on one calibration page) so that the pages carry paragraphs with roles AND tables with rows, columns and cells - with the
    # plain seeded heads of rounds 4-5 this tool's pages had no table at all
    import e2e_oracle_eval as ev

    sds = ev.state_dicts()
    an.text_detector.model.load_state_dict(sds["det"])
    an.text_recognizer.model.load_state_dict(sds["rec"])
    an.layout.layout_parser.model.load_state_dict(sds["lay"])
    an.layout.table_structure_recognizer.model.load_state_dict(sds["tab"])
    pages = ev.pages(n)
    every = [an.text_detector.model, an.layout.layout_parser.model, an.layout.table_structure_recognizer.model, an.text_recognizer.model]
    for m in every:
        m.set_conv_split(0)  # the yardstick: EXACT fp32 in all four nets (the models' default is the fp16 split since round 4)
    base = [r.model_dump() for r in an.serve(pages)]
    code = int(os.environ.get("SPLIT", "16"))
    nets = every[:3]
    if os.environ.get("ALL") == "1":
        nets = every
    for m in nets:
        m.set_conv_split(code)
    split = [r.model_dump() for r in an.serve(pages)]
    arms = []
    if os.environ.get("ROUTES_CONTROL") == "1":
        # the same split arithmetic on the round-4 routing: no A-stationary kernel, LayerNorm as its own launch, fp32
        # activations everywhere - what the round-5 routes (one of which, the planes under a bound, changes a scale) add
        from yomitoku_amd import _lib
        for k, v in (("astat", 0), ("parseq_no_ln_fusion", 1), ("act_planes", 0)):
            _lib.debug_option(k, v)
        try:
            arms.append(("split_round4_routing_vs_fp32", [r.model_dump() for r in an.serve(pages)]))
        finally:
            for k, v in (("astat", 1), ("parseq_no_ln_fusion", 0), ("act_planes", 1)):
                _lib.debug_option(k, v)
    for m in nets:
        m.set_conv_split(0)
    again = [r.model_dump() for r in an.serve(pages)]
    arms = [("split_det_layout_table_vs_fp32" if len(nets) == 3 else "split_all_four_nets_vs_fp32", split), ("fp32_repeat_vs_fp32", again)] + arms
    if os.environ.get("CONTROL") == "1":
        # the yardstick: EXACT fp32 products summed in another order (every convolution through the split-K kernel, which adds
        # the K tiles of a row in four interleaved chains) - what any second fp32 implementation differs by
        from yomitoku_amd import _lib
        _lib.debug_option("splitk_force", 0)
        try:
            arms.append(("exact_fp32_other_summation_order_vs_fp32", [r.model_dump() for r in an.serve(pages)]))
        finally:
            _lib.debug_option("splitk_force", -1)
    out = {"conv_split": code, "nets": "all four" if len(nets) == 4 else "detector + layout + table", "pages": n, "words": sum(len(d["words"]) for d in base), "tables": sum(len(d["tables"]) for d in base), "cells": sum(len(t["cells"]) for d in base for t in d["tables"]),
           "paragraphs": sum(len(d["paragraphs"]) for d in base)}
    for label, other in arms:
        st = {"discrete": 0, "leaves": 0, "float_rel": 0.0}
        pages_diff = 0
        for a, b in zip(other, base):
            before = st["discrete"]
            schema_diff(a, b, st)
            pages_diff += st["discrete"] > before
        out[label] = {"discrete_leaves": st["leaves"], "discrete_leaves_differing": st["discrete"], "pages_with_a_difference": pages_diff,
                      "max_rel_score_diff": st["float_rel"]}
    print(json.dumps(out))
    an.close()


if __name__ == "__main__":
    main()

"""Diagnosis aid: the whole-page producer-record check (tests/test_conv_split_gpu.py) at amax_check level 2, which names
every launch whose max|x| record lies below the measured maximum of its input on stderr."""
import sys

import torch

sys.path.insert(0, ".")
from yomitoku_amd import DocumentAnalyzer, _lib  # noqa: E402
from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_page_with_truth  # noqa: E402
from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict  # noqa: E402

lite = {"ocr": {"text_detector": {"from_pretrained": False},
                "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True, "batch_bucketing": True, "source_downscale": True}},
        "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}}}
an = DocumentAnalyzer(configs=lite, device="cuda:0")
an.text_detector.model.load_state_dict(dbnet_state_dict(1234, out_bias=-2.0))
an.text_recognizer.model.load_state_dict(parseq_state_dict(1235, eos_bias=6.0))
an.layout.layout_parser.model.load_state_dict(rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0))
an.layout.table_structure_recognizer.model.load_state_dict(rtdetr_state_dict(1243, num_classes=3, score_bias=-1.0))
pages = [synthetic_page_with_truth(100 + i, *((1600, 1200) if i % 2 else (1200, 1600)))[0] for i in range(8)]
mode = sys.argv[1] if len(sys.argv) > 1 else "serve"
for level in (2,):
    before = _lib.amax_check_counters()
    _lib.debug_option("amax_check", level)
    try:
        if mode == "serve":
            res = an.serve(pages, wave=8, in_flight=1)
        else:  # one net at a time, one stream
            from yomitoku_amd import imaging

            devp = [imaging.page_to_device(p, "cuda:0") for p in pages]
            for name, fn in (("detector", lambda: an.text_detector.forward_pages(devp)), ("layout", lambda: an.layout.layout_parser.forward_pages(devp))):
                b0 = _lib.amax_check_counters()
                fn()
                torch.cuda.synchronize()
                print(name, [a - b for a, b in zip(_lib.amax_check_counters(), b0)], file=sys.stderr)
    finally:
        _lib.debug_option("amax_check", 0)
    print("level", level, "checked/below/loose/worst", [a - b for a, b in zip(_lib.amax_check_counters(), before)], file=sys.stderr)
an.close()

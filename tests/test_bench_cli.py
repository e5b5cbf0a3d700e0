"""bench.py without a GPU: the module imports, its workloads / presets line up with the module catalogs and
constructor signatures, and running it on a box with no HIP device fails loudly (no CPU fallback)."""
import inspect
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_presets_match_catalogs_and_signatures():
    sys.path.insert(0, ROOT)
    import bench
    from yomitoku_amd.layout_parser import LayoutParser
    from yomitoku_amd.table_structure_recognizer import TableStructureRecognizer
    from yomitoku_amd.text_detector import TextDetector
    from yomitoku_amd.text_recognizer import TextRecognizer

    for name in bench.REC_PRESETS:
        assert name in TextRecognizer.model_catalog.list_model()
    lite = bench.LITE_CONFIGS
    for cls, kwargs in ((TextDetector, lite["ocr"]["text_detector"]), (TextRecognizer, lite["ocr"]["text_recognizer"]),
                        (LayoutParser, lite["layout_analyzer"]["layout_parser"]),
                        (TableStructureRecognizer, lite["layout_analyzer"]["table_structure_recognizer"])):
        params = inspect.signature(cls.__init__).parameters
        assert set(kwargs) <= set(params), (cls.__name__, set(kwargs) - set(params))
    # the --lite recogniser switches of the reference CLI (cli/main.py:505-520)
    rec = lite["ocr"]["text_recognizer"]
    assert rec["model_name"] == "parseq-tiny-dynw-v4" and rec["dynamic_width"] and rec["batch_bucketing"] and rec["source_downscale"]
    assert set(bench.CKPT) == {"det", "rec", "lay", "tab"}
    for name, cfgs in bench.MODEL_SETS.items():
        assert name in bench.REC_CKPT_OF_SET
        assert set(cfgs["ocr"]["text_recognizer"]) <= set(inspect.signature(TextRecognizer.__init__).parameters)


def test_bench_refuses_to_run_without_a_hip_device():
    import torch

    if torch.cuda.is_available():
        return  # the GPU tiers run the real thing
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "HIP device" in (out.stderr + out.stdout)

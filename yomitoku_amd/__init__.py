"""yomitoku_amd - MI355X-native DocumentAnalyzer inference path with yomitoku's Python API.

    from yomitoku_amd import DocumentAnalyzer, OCR, LayoutAnalyzer
    analyzer = DocumentAnalyzer(configs={...}, device="cuda")
    results, ocr_vis, layout_vis = analyzer(img_bgr_uint8)

Importing the package does not load the HIP library; the first module construction does
(`yomitoku_amd._lib.load()` raises if libymk_hip.so has not been built - there is no CPU fallback).
"""

import os as _os

# DocumentAnalyzer.serve keeps three compute streams, a copy stream and the default stream busy at once; the HIP runtime maps
# streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise.  The runtime reads the
# variable when it initialises (the process's first HIP call), so it is set at import time - unless the user already chose.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

__version__ = "0.1.0"

__all__ = ["DocumentAnalyzer", "OCR", "LayoutAnalyzer", "TextDetector", "TextRecognizer", "LayoutParser",
           "TableStructureRecognizer"]


def __getattr__(name):
    if name in ("DocumentAnalyzer", "OCR", "LayoutAnalyzer"):
        from . import document_analyzer

        return getattr(document_analyzer, name)
    if name == "TextDetector":
        from .text_detector import TextDetector

        return TextDetector
    if name == "TextRecognizer":
        from .text_recognizer import TextRecognizer

        return TextRecognizer
    if name == "LayoutParser":
        from .layout_parser import LayoutParser

        return LayoutParser
    if name == "TableStructureRecognizer":
        from .table_structure_recognizer import TableStructureRecognizer

        return TableStructureRecognizer
    raise AttributeError(name)

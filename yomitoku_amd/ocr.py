"""`from yomitoku.ocr import OCR` of the reference (ocr.py:6-63) - the class lives in document_analyzer.py here."""
from .document_analyzer import OCR, ocr_aggregate  # noqa: F401

__all__ = ["OCR", "ocr_aggregate"]

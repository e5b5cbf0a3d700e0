"""`from yomitoku.layout_analyzer import LayoutAnalyzer` of the reference (layout_analyzer.py:7-49) - the class lives
in document_analyzer.py here."""
from .document_analyzer import LayoutAnalyzer  # noqa: F401

__all__ = ["LayoutAnalyzer"]

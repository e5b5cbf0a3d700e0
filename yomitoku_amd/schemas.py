"""Result types of the path (reference schemas/document_analyzer.py:9-254): same class names, field
names, types and validation (extra fields forbidden, validate on assignment)."""

from __future__ import annotations

from typing import Any, Dict, List, Optional

from pydantic import Field, conlist

from .base import BaseSchema

Box = conlist(int, min_length=4, max_length=4)
Quad = conlist(conlist(int, min_length=2, max_length=2), min_length=4, max_length=4)


class Element(BaseSchema):
    id: Optional[str]
    box: Box
    score: float
    role: Optional[str]
    contents: Optional[str]


class ParagraphSchema(BaseSchema):
    box: Box
    contents: Optional[str]
    direction: Optional[str]
    order: Optional[int]
    role: Optional[str]


class TableCellSchema(BaseSchema):
    col: int
    row: int
    col_span: int
    row_span: int
    box: Box
    contents: Optional[str]


class TableLineSchema(BaseSchema):
    box: Box
    score: float


class TableStructureRecognizerSchema(BaseSchema):
    box: Box
    n_row: int
    n_col: int
    rows: List[TableLineSchema]
    cols: List[TableLineSchema]
    spans: List[TableLineSchema]
    cells: List[TableCellSchema]
    order: int


class LayoutAnalyzerSchema(BaseSchema):
    paragraphs: List[Element]
    tables: List[TableStructureRecognizerSchema]
    figures: List[Element]


class WordPrediction(BaseSchema):
    points: Quad
    content: str
    direction: str
    rec_score: float
    det_score: float


class TextDetectorSchema(BaseSchema):
    points: List[Quad]
    scores: List[float]


class OCRSchema(BaseSchema):
    words: List[WordPrediction]


class LayoutParserSchema(BaseSchema):
    paragraphs: List[Element]
    tables: List[Element]
    figures: List[Element]


class FigureSchema(BaseSchema):
    box: Box
    order: Optional[int]
    paragraphs: List[ParagraphSchema]
    direction: Optional[str]
    figure_path: Optional[str] = None


class DocumentAnalyzerSchema(BaseSchema):
    paragraphs: List[ParagraphSchema]
    tables: List[TableStructureRecognizerSchema]
    words: List[WordPrediction]
    figures: List[FigureSchema]

    # exporters (reference schemas/document_analyzer.py:221-231 -> export/*.py)
    def to_html(self, out_path: str, **kwargs):
        from .export import export_html

        return export_html(self, out_path, **kwargs)

    def to_markdown(self, out_path: str, **kwargs):
        from .export import export_markdown

        return export_markdown(self, out_path, **kwargs)

    def to_csv(self, out_path: str, **kwargs):
        from .export import export_csv

        return export_csv(self, out_path, **kwargs)

    def to_json(self, out_path: str, **kwargs):
        from .export import export_json

        return export_json(self, out_path, **kwargs)


class TextRecognizerSchema(BaseSchema):
    contents: List[str]
    directions: List[str]
    scores: List[float]
    points: List[Quad]


# ---- table cell detector (reference schemas/table_semantic_parser.py:62-145: the model-stage records)
class CellSchema(BaseSchema):
    meta: Dict[str, Any] = Field(default_factory=dict)
    contents: Optional[str]
    role: Optional[str]
    id: Optional[str]
    box: Box
    row: Optional[int]
    col: Optional[int]
    row_span: Optional[int]
    col_span: Optional[int]


class RegionSchema(BaseSchema):
    id: Optional[str] = None
    box: Box
    role: str
    score: float = 1.0


class TableDetectorSchema(BaseSchema):
    id: Optional[str]
    box: Box
    role: Optional[str]
    cells: List[CellSchema]
    kv_regions: List[RegionSchema] = Field(default_factory=list)
    grid_regions: List[RegionSchema] = Field(default_factory=list)

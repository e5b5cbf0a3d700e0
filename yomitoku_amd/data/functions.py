"""load_image / load_pdf with the reference's contract (data/functions.py:19-193): same accepted formats, same
exceptions and messages, pages as uint8 H x W x 3 BGR arrays.  Pillow decodes images (as in the reference); PDF
rasterisation needs pypdfium2, which is an optional dependency here - without it load_pdf raises ImportError naming the
package (there is no fallback rasteriser)."""

from __future__ import annotations

import contextlib
from pathlib import Path

import numpy as np

from ..base import logger

SUPPORT_INPUT_FORMAT = ["jpg", "jpeg", "png", "bmp", "tiff", "tif", "pdf"]
MIN_IMAGE_SIZE = 32
WARNING_IMAGE_SIZE = 720


def validate_image(img: np.ndarray):
    h, w = img.shape[:2]
    if h < MIN_IMAGE_SIZE or w < MIN_IMAGE_SIZE:
        raise ValueError("Image size is too small.")
    if min(h, w) < WARNING_IMAGE_SIZE:
        logger.warning(
            "The image size is small, which may result in reduced OCR accuracy. The process will continue, but it is "
            "recommended to input images with a minimum size of 720 pixels on the shorter side."
        )


def _check_path(path, want_pdf: bool) -> Path:
    path = Path(path)
    if not path.exists():
        raise FileNotFoundError(f"File not found: {path}")
    ext = path.suffix[1:].lower()
    if ext not in SUPPORT_INPUT_FORMAT:
        raise ValueError(f"Unsupported image format. Supported formats are {SUPPORT_INPUT_FORMAT}")
    if ext == "pdf" and not want_pdf:
        raise ValueError("PDF file is not supported by load_image(). Use load_pdf() instead.")
    if ext != "pdf" and want_pdf:
        raise ValueError("image file is not supported by load_pdf(). Use load_image() instead.")
    return path


def _decode(path: Path):
    """Pillow handle of an image file; anything Pillow cannot open is the reference's ValueError."""
    from PIL import Image

    try:
        return Image.open(path)
    except Exception:
        raise ValueError("Invalid image data.")


def _page_from_frame(frame) -> np.ndarray:
    rgb = np.array(frame.convert("RGB"))  # a writable copy, as the reference hands out
    validate_image(rgb)
    return rgb[:, :, ::-1]  # BGR view, as every module expects its input


def load_image(image_path: str):
    """Image file -> list of BGR pages: one, or every frame of a multi-page TIFF (contract of data/functions.py:33-79)."""
    from PIL import ImageSequence

    path = _check_path(image_path, want_pdf=False)
    with _decode(path) as img:
        multi_frame = path.suffix.lower() in (".tif", ".tiff")
        return [_page_from_frame(f) for f in (ImageSequence.Iterator(img) if multi_frame else (img,))]


def _pdfium():
    try:
        import pypdfium2
    except ImportError as exc:
        raise ImportError("load_pdf needs the pypdfium2 package (PDF rasterisation at 200 dpi, data/functions.py:91-101); "
                          "it is not installed and there is no fallback rasteriser") from exc
    return pypdfium2


class PdfPageIterator:
    """Pages of a PDF rendered on demand, one at a time (contract of data/functions.py:82-160): len(), integer / slice
    indexing, iteration; a page is rasterised at `dpi` and returned as a BGR array.  The document is opened per request and
    closed when the request is served, so a thousand-page file never sits in memory."""

    def __init__(self, pdf_path, dpi: int = 200):
        self._pdf_path = Path(pdf_path)
        self._zoom = dpi / 72
        _pdfium()
        with self._document() as doc:
            self.total_pages = len(doc)

    @contextlib.contextmanager
    def _document(self):
        try:
            doc = _pdfium().PdfDocument(self._pdf_path)
        except Exception as e:
            raise ValueError(f"Failed to open the PDF file: {self._pdf_path}") from e
        try:
            yield doc
        finally:
            doc.close()

    def _rasterise(self, numbers):
        """Generator over the BGR rasters of the given page numbers (one open document for the whole request)."""
        with self._document() as doc:
            for i in numbers:
                yield np.array(doc[i].render(scale=self._zoom).to_pil().convert("RGB"))[:, :, ::-1]

    def __len__(self):
        return self.total_pages

    def __iter__(self):
        return self._rasterise(range(self.total_pages))

    def __getitem__(self, index):
        if isinstance(index, slice):
            return list(self._rasterise(range(*index.indices(self.total_pages))))
        if not isinstance(index, int):
            raise TypeError(f"indices must be integers or slices, not {type(index).__name__}")
        number = index + self.total_pages if index < 0 else index
        if not 0 <= number < self.total_pages:
            raise IndexError(f"page index {index} out of range")
        (page,) = self._rasterise((number,))  # exhausted: the document is closed again
        return page


def load_pdf(pdf_path: str, dpi=200) -> PdfPageIterator:
    """PDF -> lazy iterator of BGR pages rendered at `dpi` (data/functions.py:163-193)."""
    return PdfPageIterator(_check_path(pdf_path, want_pdf=True), dpi=dpi)

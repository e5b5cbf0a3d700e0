"""load_image / load_pdf with the reference's contract (data/functions.py:19-193): same accepted formats, same
exceptions and messages, pages as uint8 H x W x 3 BGR arrays.  Pillow decodes images (as in the reference); PDF
rasterisation needs pypdfium2, which is an optional dependency here - without it load_pdf raises ImportError naming the
package (there is no fallback rasteriser)."""

from __future__ import annotations

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


def load_image(image_path: str):
    """Open an image file -> list of BGR pages (one, or every frame of a multi-page TIFF), data/functions.py:33-79."""
    from PIL import Image

    path = _check_path(image_path, want_pdf=False)
    ext = path.suffix[1:].lower()
    try:
        img = Image.open(path)
    except Exception:
        raise ValueError("Invalid image data.")
    pages = []
    if ext in ("tif", "tiff"):
        try:
            while True:
                arr = np.array(img.copy().convert("RGB"))
                validate_image(arr)
                pages.append(arr[:, :, ::-1])
                img.seek(img.tell() + 1)
        except EOFError:
            pass
    else:
        arr = np.array(img.convert("RGB"))
        validate_image(arr)
        pages.append(arr[:, :, ::-1])
    return pages


def _pdfium():
    try:
        import pypdfium2
    except ImportError as exc:
        raise ImportError("load_pdf needs the pypdfium2 package (PDF rasterisation at 200 dpi, data/functions.py:91-101); "
                          "it is not installed and there is no fallback rasteriser") from exc
    return pypdfium2


class PdfPageIterator:
    """Lazy page-by-page rendering of a PDF (data/functions.py:82-160): len(), integer / slice indexing, iteration;
    every page is rendered at `dpi` and returned as a BGR array."""

    def __init__(self, pdf_path, dpi: int = 200):
        self._pdf_path = Path(pdf_path)
        self._dpi = dpi
        pdfium = _pdfium()
        try:
            doc = pdfium.PdfDocument(self._pdf_path)
            self.total_pages = len(doc)
            doc.close()
        except Exception as e:
            raise ValueError(f"Failed to open the PDF file: {pdf_path}") from e

    def __len__(self):
        return self.total_pages

    def _open(self):
        try:
            return _pdfium().PdfDocument(self._pdf_path)
        except Exception as e:
            raise ValueError(f"Failed to open the PDF file: {self._pdf_path}") from e

    def _render_page(self, doc, index: int) -> np.ndarray:
        bitmap = doc[index].render(scale=self._dpi / 72)
        return np.array(bitmap.to_pil().convert("RGB"))[:, :, ::-1]

    def __getitem__(self, index):
        if isinstance(index, slice):
            doc = self._open()
            try:
                return [self._render_page(doc, i) for i in range(*index.indices(self.total_pages))]
            finally:
                doc.close()
        if isinstance(index, int):
            if index < 0:
                index += self.total_pages
            if not (0 <= index < self.total_pages):
                raise IndexError(f"page index {index} out of range")
            doc = self._open()
            try:
                return self._render_page(doc, index)
            finally:
                doc.close()
        raise TypeError(f"indices must be integers or slices, not {type(index).__name__}")

    def __iter__(self):
        doc = self._open()
        try:
            for i in range(self.total_pages):
                yield self._render_page(doc, i)
        finally:
            doc.close()


def load_pdf(pdf_path: str, dpi=200) -> PdfPageIterator:
    """PDF -> lazy iterator of BGR pages rendered at `dpi` (data/functions.py:163-193)."""
    return PdfPageIterator(_check_path(pdf_path, want_pdf=True), dpi=dpi)

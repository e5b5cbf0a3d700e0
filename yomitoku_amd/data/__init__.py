"""Page ingestion - the step before the path (reference data/functions.py:33-193).

`load_image` / `load_pdf` keep the reference contract (lists / lazy iterators of uint8 H x W x 3 BGR pages, the same
errors).  The MI355X part is `PageStager`: decoded pages go through a small ring of PINNED host buffers and are copied to
HBM asynchronously on a dedicated copy stream, so page i + 1 uploads (5.8 MB at 1600 x 1200, ~0.1 ms of PCIe time) while
page i computes; consumers wait on a HIP event, never on the host.  `stream_pages` strings file decoding (a background
thread), staging and the hand-over to DocumentAnalyzer.analyze_pages together.
"""

from .functions import MIN_IMAGE_SIZE, SUPPORT_INPUT_FORMAT, WARNING_IMAGE_SIZE, PdfPageIterator, load_image, load_pdf, validate_image
from .staging import PageStager, stream_pages

__all__ = ["load_image", "load_pdf", "PdfPageIterator", "validate_image", "PageStager", "stream_pages", "MIN_IMAGE_SIZE",
           "WARNING_IMAGE_SIZE", "SUPPORT_INPUT_FORMAT"]

"""Searchable PDF: every page image with its recognised words as an invisible text layer.

Mirror of the reference's `utils/searchable_pdf.py:74 create_searchable_pdf` (called by cli/main.py:73 and :264): same
signature, same choice and order of words (the words of every paragraph / table cell / figure paragraph, containers in reading
order, words inside a container by its direction; a word in no container is left out, a word in two is written twice), same
placement rule (the font size out of [0.5, 1.0) x the box height whose text width comes closest to the box width; vertical
words converted to full-width forms and written one character at a time, turned by -90 degrees).

The reference draws through reportlab with an embedded TrueType face and converts half-width text with jaconv.  Neither is
a dependency here: this module writes the PDF itself.

  * the image: a JPEG (the reference's three quality presets) as a DCTDecode XObject filling the page, one pixel = one point;
  * the text: rendering mode 3 (neither filled nor stroked - the reference fills with alpha 0), a composite font that is not
    embedded (no glyph is ever drawn), Identity-H codes = the characters' BMP code points, and a ToUnicode CMap that maps
    every code to itself - so search / copy give back exactly the recognised strings in any viewer, whichever face it
    substitutes.  A character beyond the BMP (a CJK Extension B kanji, say) has no 2-byte code point: it gets a code from the
    surrogate block U+D800..DFFF - which no text uses - in order of first appearance in the document, and a `bfchar` entry
    that maps that one code to the character's UTF-16 pair (two codes that each map to a lone surrogate, which is what
    writing the UTF-16 code units would produce, copy as garbage);
  * widths: one em for everything from U+1100 up except the half-width forms, half an em below - the model `string_width`
    uses to choose the font size and the /W array the viewer uses to place the selection, so the two agree (the reference
    asks the TrueType face for its advance widths; with an invisible layer only the selection rectangle depends on them).

One deliberate difference: with image_quality "middle" / "low" the reference shrinks the image and the page but keeps the word
coordinates of the full-size page (utils/searchable_pdf.py:103-113 against :189-226), which leaves the text layer beside the
picture; here the coordinates are scaled with the image.  `font_path` is accepted and not used.
"""

from __future__ import annotations

import unicodedata
import zlib
from io import BytesIO
from typing import List, Optional, Sequence

import numpy as np
from PIL import Image

from ..geometry import is_contained

IMAGE_QUALITY_PRESETS = {
    "high": {"max_long_side": None, "jpeg_quality": 85},
    "middle": {"max_long_side": 2000, "jpeg_quality": 80},
    "low": {"max_long_side": 1500, "jpeg_quality": 60},
}

FONT_NAME = "YmkInvisibleText"


# ------------------------------------------------------------------------------------------------ text conversion
_VOICED, _SEMI_VOICED = "ﾞ", "ﾟ"


def h2z(text: str) -> str:
    """Half-width -> full-width forms of ASCII, digits and katakana (what the reference asks of jaconv.h2z with kana, ascii
    and digit all on): U+0021..U+007E -> U+FF01..U+FF5E, space -> U+3000, half-width katakana and its punctuation
    (U+FF61..U+FF9F) -> the full-width letter, taking a following (semi-)voiced mark into the letter where one exists."""
    out = []
    i, n = 0, len(text)
    while i < n:
        ch = text[i]
        cp = ord(ch)
        if 0x21 <= cp <= 0x7E:
            out.append(chr(cp + 0xFEE0))
        elif ch == " ":
            out.append("　")
        elif 0xFF61 <= cp <= 0xFF9D:
            if i + 1 < n and text[i + 1] in (_VOICED, _SEMI_VOICED):
                joined = unicodedata.normalize("NFKC", ch + text[i + 1])
                if len(joined) == 1:
                    out.append(joined)
                    i += 2
                    continue
            out.append(unicodedata.normalize("NFKC", ch))
        elif ch == _VOICED:
            out.append("゛")
        elif ch == _SEMI_VOICED:
            out.append("゜")
        else:
            out.append(ch)
        i += 1
    return "".join(out)


_FULL_WIDTH_EXTRA = str.maketrans({"¥": "￥", "·": "・", " ": "　"})


def to_full_width(text: str) -> str:
    """utils/searchable_pdf.py:59: h2z, then the yen sign, the middle dot and any space left."""
    return h2z(text).translate(_FULL_WIDTH_EXTRA)


# ------------------------------------------------------------------------------------------------ widths and font size
def _is_narrow(cp: int) -> bool:
    return cp < 0x1100 or 0xFF61 <= cp <= 0xFFDC or 0xFFE8 <= cp <= 0xFFEE


def string_width(text: str, font_size: float) -> float:
    """Advance width of `text`: half an em per narrow character, one em per wide one (the /W array below says the same)."""
    half_ems = sum(1 if _is_narrow(ord(ch)) else 2 for ch in text)
    return half_ems * 0.5 * font_size


_RATES = np.arange(0.5, 1.0, 0.01)


def calc_font_size(content: str, bbox_height, bbox_width, width_of=string_width):
    """utils/searchable_pdf.py:42: of the sizes 0.50, 0.51, ... 0.99 x the box height, the first whose text width is closest
    to the box width."""
    best, best_diff = None, np.inf
    for rate in _RATES:
        font_size = bbox_height * rate
        diff = abs(width_of(content, font_size) - bbox_width)
        if diff < best_diff:
            best_diff, best = diff, font_size
    return best


def poly2rect(points):
    """Axis-aligned hull of a quadrangle, coordinates truncated to integers first (utils/searchable_pdf.py:29)."""
    pts = np.array(points, dtype=int)
    return [pts[:, 0].min(), pts[:, 1].min(), pts[:, 0].max(), pts[:, 1].max()]


# ------------------------------------------------------------------------------------------------ which words, where
def words_in_reading_order(doc) -> list:
    """utils/searchable_pdf.py:122-187: paragraphs, table cells (by row, column) and figure paragraphs as containers sorted by
    (order, sub-order); a container takes the words of which more than 0.7 lies inside it, sorted right-to-left then
    top-to-bottom when it is vertical, top-to-bottom then left-to-right otherwise."""
    containers = []
    for p in doc.paragraphs:
        containers.append((p.order, 0, p.box, p.direction))
    for t in doc.tables:
        for cell in t.cells:
            containers.append((t.order, (cell.row, cell.col), cell.box, "horizontal"))
    for f in doc.figures:
        for k, p in enumerate(f.paragraphs):
            containers.append((f.order, k, p.box, p.direction))
    containers.sort(key=lambda c: (c[0], c[1]))
    rects = [poly2rect(w.points) for w in doc.words]
    out = []
    for _, _, box, direction in containers:
        inside = [k for k, r in enumerate(rects) if is_contained(box, r, 0.7)]
        if direction == "vertical":
            inside.sort(key=lambda k: (-rects[k][0], rects[k][1]))
        else:
            inside.sort(key=lambda k: (rects[k][1], rects[k][0]))
        out.extend(doc.words[k] for k in inside)
    return out


def text_layer(doc, page_height, width_of=string_width) -> list:
    """The drawing operations of one page's text layer, in order: ("font", size) and ("text", a, b, c, d, e, f, string) with
    the text matrix in PDF user space (origin bottom-left, y up; `page_height` flips the image coordinates).
    utils/searchable_pdf.py:193-231."""
    ops = []
    for word in words_in_reading_order(doc):
        text = word.content
        x1, y1, x2, y2 = poly2rect(word.points)
        bbox_height, bbox_width = y2 - y1, x2 - x1
        if word.direction == "vertical":
            text = to_full_width(text)
            font_size = calc_font_size(text, bbox_width, bbox_height, width_of)
        else:
            font_size = calc_font_size(text, bbox_height, bbox_width, width_of)
        if not font_size:
            continue
        ops.append(("font", float(font_size)))
        if word.direction == "vertical":
            base_y = page_height - y1
            char_height = bbox_height / len(text) if text else 0
            for j, ch in enumerate(text):
                char_x = x1 + (bbox_width - font_size) / 2
                char_y = base_y - (j * char_height) - char_height / 2
                ops.append(("text", 0.0, -1.0, 1.0, 0.0, float(char_x), float(char_y + font_size / 2), ch))
        else:
            base_y = page_height - y2 + (bbox_height - font_size) * 0.5
            ops.append(("text", 1.0, 0.0, 0.0, 1.0, float(x1), float(base_y), text))
    return ops


# ------------------------------------------------------------------------------------------------ the file
def _num(v: float) -> str:
    s = f"{v:.4f}".rstrip("0").rstrip(".")
    return "0" if s in ("", "-0") else s


SUPPLEMENTARY_BASE, SUPPLEMENTARY_SLOTS = 0xD800, 0x800  # the codes handed to characters beyond the BMP


def _hex_codes(text: str, supplementary: dict) -> str:
    """The 2-byte codes of `text` as a hex string; `supplementary` (character -> code) grows as new ones appear."""
    out = []
    for ch in text:
        cp = ord(ch)
        if 0xD800 <= cp <= 0xDFFF:  # a lone surrogate in the input: not a character
            cp = 0xFFFD
        elif cp > 0xFFFF:
            if ch not in supplementary:
                if len(supplementary) >= SUPPLEMENTARY_SLOTS:
                    raise ValueError(f"more than {SUPPLEMENTARY_SLOTS} distinct characters beyond the BMP in one PDF")
                supplementary[ch] = SUPPLEMENTARY_BASE + len(supplementary)
            cp = supplementary[ch]
        out.append(f"{cp:04X}")
    return "".join(out)


def _to_unicode_cmap(supplementary: dict) -> bytes:
    rows = [f"<{hi:02X}00> <{hi:02X}FF> <{hi:02X}00>" for hi in range(256) if not 0xD8 <= hi <= 0xDF]
    blocks = []
    for k in range(0, len(rows), 100):
        part = rows[k : k + 100]
        blocks.append(f"{len(part)} beginbfrange\n" + "\n".join(part) + "\nendbfrange")
    chars = [f"<{code:04X}> <{ch.encode('utf-16-be').hex().upper()}>" for ch, code in sorted(supplementary.items(), key=lambda kv: kv[1])]
    for k in range(0, len(chars), 100):
        part = chars[k : k + 100]
        blocks.append(f"{len(part)} beginbfchar\n" + "\n".join(part) + "\nendbfchar")
    return ("/CIDInit /ProcSet findresource begin\n12 dict begin\nbegincmap\n"
            "/CIDSystemInfo << /Registry (Adobe) /Ordering (UCS) /Supplement 0 >> def\n"
            "/CMapName /Adobe-Identity-UCS def\n/CMapType 2 def\n"
            "1 begincodespacerange\n<0000> <FFFF>\nendcodespacerange\n" + "\n".join(blocks) +
            "\nendcmap\nCMapName currentdict /CMap defineresource pop\nend\nend\n").encode("ascii")


class _PdfFile:
    """Indirect objects in the order they are numbered, a classic cross-reference table, nothing incremental."""

    def __init__(self):
        self._bodies: List[Optional[bytes]] = []

    def reserve(self) -> int:
        self._bodies.append(None)
        return len(self._bodies)

    def put(self, num: int, body: bytes):
        self._bodies[num - 1] = body

    def add(self, body: bytes) -> int:
        num = self.reserve()
        self.put(num, body)
        return num

    def add_stream(self, head: str, data: bytes, compress: bool) -> int:
        if compress:
            data = zlib.compress(data, 6)
            head += " /Filter /FlateDecode"
        return self.add(f"<< {head} /Length {len(data)} >>\nstream\n".encode("ascii") + data + b"\nendstream")

    def tobytes(self, root: int) -> bytes:
        out = BytesIO()
        out.write(b"%PDF-1.5\n%\xe2\xe3\xcf\xd3\n")
        offsets = []
        for k, body in enumerate(self._bodies):
            if body is None:
                raise RuntimeError(f"object {k + 1} was reserved and never written")
            offsets.append(out.tell())
            out.write(f"{k + 1} 0 obj\n".encode("ascii") + body + b"\nendobj\n")
        xref = out.tell()
        out.write(f"xref\n0 {len(offsets) + 1}\n".encode("ascii"))
        out.write(b"0000000000 65535 f \n")
        for off in offsets:
            out.write(f"{off:010d} 00000 n \n".encode("ascii"))
        out.write(f"trailer\n<< /Size {len(offsets) + 1} /Root {root} 0 R >>\nstartxref\n{xref}\n%%EOF\n".encode("ascii"))
        return out.getvalue()


def _font_objects(pdf: _PdfFile, font: int, supplementary: dict):
    """The composite font into the reserved object `font` - written after the pages, when every character beyond the BMP has
    its code."""
    descriptor = pdf.add((f"<< /Type /FontDescriptor /FontName /{FONT_NAME} /Flags 4 /FontBBox [0 -120 1000 880] /ItalicAngle 0 "
                          "/Ascent 880 /Descent -120 /CapHeight 880 /StemV 80 >>").encode("ascii"))
    # /DW 1000: one em (the codes of the surrogate block included: characters beyond the BMP are wide); the narrow ranges of
    # _is_narrow at half an em
    cid = pdf.add((f"<< /Type /Font /Subtype /CIDFontType2 /BaseFont /{FONT_NAME} "
                   "/CIDSystemInfo << /Registry (Adobe) /Ordering (Identity) /Supplement 0 >> "
                   f"/FontDescriptor {descriptor} 0 R /DW 1000 /W [0 4351 500 65377 65500 500 65512 65518 500] >>").encode("ascii"))
    to_unicode = pdf.add_stream("", _to_unicode_cmap(supplementary), compress=True)
    pdf.put(font, (f"<< /Type /Font /Subtype /Type0 /BaseFont /{FONT_NAME} /Encoding /Identity-H "
                   f"/DescendantFonts [{cid} 0 R] /ToUnicode {to_unicode} 0 R >>").encode("ascii"))


def _content_stream(width: int, height: int, ops: Sequence, supplementary: dict) -> bytes:
    lines = [f"q {width} 0 0 {height} 0 0 cm /Im0 Do Q", "BT", "3 Tr"]
    for op in ops:
        if op[0] == "font":
            lines.append(f"/F1 {_num(op[1])} Tf")
        elif op[7]:
            lines.append(" ".join(_num(v) for v in op[1:7]) + f" Tm <{_hex_codes(op[7], supplementary)}> Tj")
    lines.append("ET")
    return ("\n".join(lines) + "\n").encode("ascii")


def _as_pil(image) -> Image.Image:
    """A PIL image as the reference takes it, or - as every module of this package takes pages - a BGR / grey uint8 array."""
    if isinstance(image, Image.Image):
        return image
    arr = np.asarray(image)
    if arr.ndim == 3 and arr.shape[2] >= 3:
        arr = arr[:, :, 2::-1]
    return Image.fromarray(np.ascontiguousarray(arr))


def _scaled_document(doc, scale: float):
    if scale == 1.0:
        return doc
    doc = doc.model_copy(deep=True)
    boxes = [*doc.paragraphs, *(c for t in doc.tables for c in t.cells), *(p for f in doc.figures for p in f.paragraphs)]
    for el in boxes:
        el.box = [v * scale for v in el.box]
    for w in doc.words:
        w.points = [[x * scale, y * scale] for x, y in w.points]
    return doc


def searchable_pdf_bytes(images: Sequence, docs: Sequence, image_quality: str = "high") -> bytes:
    pdf = _PdfFile()
    root, pages, font = pdf.reserve(), pdf.reserve(), pdf.reserve()
    supplementary: dict = {}
    kids = []
    preset = IMAGE_QUALITY_PRESETS.get(image_quality, IMAGE_QUALITY_PRESETS["high"])
    for image, doc in zip(images, docs):
        image = _as_pil(image)
        scale = 1.0
        if preset["max_long_side"] is not None and max(image.size) > preset["max_long_side"]:
            w, h = image.size
            scale = preset["max_long_side"] / max(w, h)
            image = image.resize((int(w * scale), int(h * scale)), Image.LANCZOS)
        grey = image.mode == "L"
        if not grey and image.mode != "RGB":
            image = image.convert("RGB")
        jpeg = BytesIO()
        image.save(jpeg, format="JPEG", quality=preset["jpeg_quality"])
        w, h = image.size
        xobject = pdf.add_stream(f"/Type /XObject /Subtype /Image /Width {w} /Height {h} /ColorSpace /Device{'Gray' if grey else 'RGB'} "
                                 "/BitsPerComponent 8 /Filter /DCTDecode", jpeg.getvalue(), compress=False)
        contents = pdf.add_stream("", _content_stream(w, h, text_layer(_scaled_document(doc, scale), h), supplementary), compress=True)
        kids.append(pdf.add((f"<< /Type /Page /Parent {pages} 0 R /MediaBox [0 0 {w} {h}] /Contents {contents} 0 R "
                             f"/Resources << /XObject << /Im0 {xobject} 0 R >> /Font << /F1 {font} 0 R >> >> >>").encode("ascii")))
    _font_objects(pdf, font, supplementary)
    pdf.put(pages, (f"<< /Type /Pages /Count {len(kids)} /Kids [" + " ".join(f"{k} 0 R" for k in kids) + "] >>").encode("ascii"))
    pdf.put(root, f"<< /Type /Catalog /Pages {pages} 0 R >>".encode("ascii"))
    return pdf.tobytes(root)


def create_searchable_pdf(images: List, docs: List, output_path: str, font_path: Optional[str] = None, image_quality: str = "high"):
    """Write `images` (PIL images; BGR arrays are taken too) with the words of `docs` (DocumentAnalyzerSchema, one per image)
    as an invisible text layer to `output_path`.  utils/searchable_pdf.py:74."""
    data = searchable_pdf_bytes(images, docs, image_quality)
    with open(output_path, "wb") as f:
        f.write(data)

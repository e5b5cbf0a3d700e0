"""Searchable PDF (reference utils/searchable_pdf.py:74): the text layer against what the reference's own
function asks its canvas to draw (tests/golden/searchable_pdf.json, written by oracle/pin_against_reference.py
pin_searchable_pdf), and the file this package writes without reportlab, read back by a small parser: cross-reference
table, pages, the JPEG, and the strings a viewer's search would see."""
import io
import json
import os
import re
import zlib

import numpy as np
import pytest
from PIL import Image

from yomitoku_amd.schemas import DocumentAnalyzerSchema, ParagraphSchema, WordPrediction
from yomitoku_amd.utils.searchable_pdf import (calc_font_size, create_searchable_pdf, h2z, searchable_pdf_bytes, string_width, text_layer,
                                               to_full_width, words_in_reading_order)

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "searchable_pdf.json")


def _golden():
    with open(GOLDEN, encoding="utf-8") as f:
        return json.load(f)


def test_text_layer_is_what_the_reference_draws():
    g = _golden()
    n_text = n_turned = 0
    for page in g["pages"]:
        doc = DocumentAnalyzerSchema(**page["doc"])
        ops = text_layer(doc, page["size"][1])
        assert len(ops) == len(page["ops"])
        for mine, ref in zip(ops, page["ops"]):
            assert mine[0] == ref[0]
            if mine[0] == "font":
                assert mine[1] == pytest.approx(ref[1], rel=0, abs=1e-9)
            else:
                assert mine[7] == ref[7]
                assert list(mine[1:7]) == pytest.approx(ref[1:7], rel=0, abs=1e-6)
                n_text += 1
                n_turned += mine[2] == -1.0
    assert n_text > 150 and n_turned > 50  # both directions are in the fixture


def test_words_outside_every_container_are_left_out_and_shared_words_repeat():
    def quad(x1, y1, x2, y2):
        return [[x1, y1], [x2, y1], [x2, y2], [x1, y2]]

    def w(box, text):
        return WordPrediction(points=quad(*box), content=text, direction="horizontal", rec_score=1.0, det_score=1.0)

    paragraphs = [ParagraphSchema(box=[0, 0, 100, 100], contents="", direction="horizontal", order=1, role=None),
                  ParagraphSchema(box=[0, 0, 100, 50], contents="", direction="horizontal", order=0, role=None)]
    words = [w([10, 60, 90, 80], "lower"), w([10, 10, 90, 30], "upper"), w([200, 200, 260, 220], "nowhere"), w([50, 12, 90, 30], "upper right")]
    doc = DocumentAnalyzerSchema(paragraphs=paragraphs, tables=[], words=words, figures=[])
    # container of order 0 (the upper half) first: its two words by (y, x); then the whole box: all three by (y, x)
    assert [x.content for x in words_in_reading_order(doc)] == ["upper", "upper right", "upper", "upper right", "lower"]


def test_full_width_forms():
    assert h2z("ABC xyz 019") == "ＡＢＣ　ｘｙｚ　０１９"
    assert h2z("!~") == "！～"
    assert h2z("ｶﾞｷﾞｸﾞｹﾞｺﾞ") == "ガギグゲゴ"
    assert h2z("ﾊﾟﾋﾟﾌﾟﾍﾟﾎﾟ") == "パピプペポ"
    assert h2z("ｳﾞ") == "ヴ"
    assert h2z("ｱﾞ") == "ア゛"  # no voiced form of that letter: the mark stays a character of its own
    assert h2z("｡｢｣､･ｰ") == "。「」、・ー"
    assert h2z("ｦｧｨｩｪｫｬｭｮｯ") == "ヲァィゥェォャュョッ"
    assert h2z("漢字かなカナ") == "漢字かなカナ"
    assert to_full_width("¥1·2 3") == "￥１・２　３"
    for text, expected in _golden()["to_full_width"]:  # the reference's to_full_width over this package's h2z
        assert to_full_width(text) == expected


def test_font_size_search():
    # four wide characters in a box 30 high: widths 4 x size; 100 wide -> size 25 = 30 x 0.8333: the closest rate is 0.83
    assert calc_font_size("請求書類", 30, 100) == pytest.approx(30 * 0.83)
    # narrow characters count half an em
    assert string_width("abcd", 10) == 20 and string_width("ｱｲ", 10) == 10 and string_width("全角", 10) == 20
    assert calc_font_size("x", 0, 50) == 0  # a flat box: the caller skips the word
    assert calc_font_size("", 40, 50) == pytest.approx(20.0)  # nothing to measure: the first rate


# ------------------------------------------------------------------------------------------------ reading the file back
def _objects(data: bytes):
    m = re.search(rb"startxref\n(\d+)\n%%EOF\n$", data)
    xref = int(m.group(1))
    assert data[xref : xref + 5] == b"xref\n"
    lines = data[xref:].split(b"\n")
    first, count = (int(v) for v in lines[1].split())
    assert first == 0 and lines[2] == b"0000000000 65535 f "
    objs = {}
    for num in range(1, count):
        entry = lines[2 + num]
        assert len(entry) == 19 and entry.endswith(b" 00000 n "), entry
        off = int(entry[:10])
        head = f"{num} 0 obj\n".encode()
        assert data[off : off + len(head)] == head, (num, data[off : off + 20])
        body_at = off + len(head)
        s = data.find(b"\nstream\n", body_at, body_at + 600)
        e = data.find(b"\nendobj\n", body_at)
        if s != -1 and s < e:
            dict_part = data[body_at:s]
            length = int(re.search(rb"/Length (\d+)", dict_part).group(1))
            raw = data[s + 8 : s + 8 + length]
            assert data[s + 8 + length : s + 8 + length + 18] == b"\nendstream\nendobj\n"
            if b"/FlateDecode" in dict_part:
                raw = zlib.decompress(raw)
            objs[num] = (dict_part, raw)
        else:
            objs[num] = (data[body_at:e], None)
    trailer = data[data.rfind(b"trailer") :]
    root = int(re.search(rb"/Root (\d+) 0 R", trailer).group(1))
    assert int(re.search(rb"/Size (\d+)", trailer).group(1)) == count
    return objs, root


def _ref(d: bytes, key: str) -> int:
    return int(re.search(rf"/{key} (\d+) 0 R".encode(), d).group(1))


def _pages(data: bytes):
    objs, root = _objects(data)
    assert data.startswith(b"%PDF-1.")
    catalog = objs[root][0]
    assert b"/Type /Catalog" in catalog
    pages = objs[_ref(catalog, "Pages")][0]
    kids = [int(k) for k in re.findall(rb"(\d+) 0 R", re.search(rb"/Kids \[(.*?)\]", pages).group(1))]
    assert int(re.search(rb"/Count (\d+)", pages).group(1)) == len(kids)
    out = []
    for k in kids:
        page = objs[k][0]
        box = [float(v) for v in re.search(rb"/MediaBox \[(.*?)\]", page).group(1).split()]
        content = objs[_ref(page, "Contents")][1].decode("ascii")
        image_dict, jpeg = objs[_ref(page, "Im0")]
        font = objs[_ref(page, "F1")][0]
        out.append({"box": box, "content": content, "image_dict": image_dict, "jpeg": jpeg, "font": font, "objs": objs})
    return out


def _strings(content: str):
    """(font size, text matrix, string) of every Tj, decoding the codes through the ToUnicode map (identity on UTF-16 units)."""
    size, out = None, []
    for line in content.split("\n"):
        m = re.fullmatch(r"/F1 (\S+) Tf", line)
        if m:
            size = float(m.group(1))
            continue
        m = re.fullmatch(r"(\S+) (\S+) (\S+) (\S+) (\S+) (\S+) Tm <([0-9A-F]*)> Tj", line)
        if m:
            out.append((size, [float(m.group(i)) for i in range(1, 7)], bytes.fromhex(m.group(7)).decode("utf-16-be")))
    return out


def test_file_reads_back_with_image_and_searchable_strings(tmp_path):
    g = _golden()
    docs = [DocumentAnalyzerSchema(**p["doc"]) for p in g["pages"][:3]]
    rng = np.random.default_rng(0)
    bgr = rng.integers(0, 255, (1400, 1000, 3), dtype=np.uint8)
    bgr[:, :, 0], bgr[:, :, 2] = 250, 5  # blue page: channel order survives the BGR -> RGB turn
    images = [Image.fromarray(np.full((1400, 1000, 3), 200, np.uint8)), bgr, Image.fromarray(np.full((1400, 1000), 90, np.uint8))]
    out = tmp_path / "out.pdf"
    create_searchable_pdf(images, docs, str(out), font_path="ignored.ttf")
    pages = _pages(out.read_bytes())
    assert len(pages) == 3
    for page, doc, ref in zip(pages, docs, g["pages"]):
        assert page["box"] == [0, 0, 1000, 1400]
        assert "3 Tr" in page["content"] and page["content"].startswith("q 1000 0 0 1400 0 0 cm /Im0 Do Q\nBT\n")
        with Image.open(io.BytesIO(page["jpeg"])) as im:
            assert im.format == "JPEG" and im.size == (1000, 1400)
            px = np.asarray(im.convert("RGB"))[700, 500]
        assert b"/Filter /DCTDecode" in page["image_dict"] and b"/Width 1000 /Height 1400" in page["image_dict"]
        # every non-empty string the reference draws, in its order, with its matrix
        want = [op for op in ref["ops"] if op[0] == "text" and op[7]]
        got = _strings(page["content"])
        assert [s for _, _, s in got] == [op[7] for op in want]
        for (size, tm, _), op in zip(got, want):
            assert tm == pytest.approx(op[1:7], abs=1e-3)
            assert size is not None and 0 < size < 400
        if doc is docs[1]:
            assert px[2] > 200 and px[0] < 60  # blue stayed blue
    assert b"/DeviceGray" in pages[2]["image_dict"] and b"/DeviceRGB" in pages[0]["image_dict"]
    font = pages[0]["font"]
    assert b"/Subtype /Type0" in font and b"/Encoding /Identity-H" in font
    cmap = pages[0]["objs"][_ref(font, "ToUnicode")][1].decode("ascii")
    assert "<3000> <30FF> <3000>" in cmap and cmap.count("beginbfrange") == 3 and "<0000> <FFFF>" in cmap
    assert "<D800>" not in cmap and "beginbfchar" not in cmap  # no character beyond the BMP in these pages: the block stays unmapped
    cid = pages[0]["objs"][int(re.search(rb"/DescendantFonts \[(\d+) 0 R\]", font).group(1))][0]
    assert b"/DW 1000" in cid and b"/W [0 4351 500 65377 65500 500 65512 65518 500]" in cid


def test_smaller_presets_scale_the_text_with_the_image():
    quad = [[400, 200], [1600, 200], [1600, 300], [400, 300]]
    doc = DocumentAnalyzerSchema(paragraphs=[ParagraphSchema(box=[300, 100, 1800, 400], contents="", direction="horizontal", order=0, role=None)],
                                 tables=[], figures=[],
                                 words=[WordPrediction(points=quad, content="見積書 No.42", direction="horizontal", rec_score=1.0, det_score=1.0)])
    image = Image.fromarray(np.full((2000, 3000, 3), 255, np.uint8))
    full = _pages(searchable_pdf_bytes([image], [doc], "high"))[0]
    low = _pages(searchable_pdf_bytes([image], [doc], "low"))[0]
    unknown = _pages(searchable_pdf_bytes([image], [doc], "no such preset"))[0]
    assert full["box"] == [0, 0, 3000, 2000] and low["box"] == [0, 0, 1500, 1000] and unknown["box"] == full["box"]
    (size_f, tm_f, s_f), = _strings(full["content"])
    (size_l, tm_l, s_l), = _strings(low["content"])
    assert s_f == s_l == "見積書 No.42"
    assert size_l == pytest.approx(size_f / 2, rel=0.03) and tm_l[4] == pytest.approx(tm_f[4] / 2, abs=1) and tm_l[5] == pytest.approx(tm_f[5] / 2, abs=1.5)
    assert len(low["jpeg"]) < len(full["jpeg"])


def test_empty_job_and_page_without_words():
    assert _pages(searchable_pdf_bytes([], [])) == []
    doc = DocumentAnalyzerSchema(paragraphs=[], tables=[], words=[], figures=[])
    page, = _pages(searchable_pdf_bytes([np.zeros((40, 60, 3), np.uint8)], [doc]))
    assert page["box"] == [0, 0, 60, 40] and _strings(page["content"]) == []


def _to_unicode_map(cmap: str) -> dict:
    """code -> string, from the bfrange (<lo> <hi> <first>) and bfchar (<code> <utf-16-be>) sections of a ToUnicode CMap."""
    table = {}
    for body in re.findall(r"beginbfrange\n(.*?)\nendbfrange", cmap, re.S):
        for lo, hi, first in re.findall(r"<([0-9A-F]{4})> <([0-9A-F]{4})> <([0-9A-F]{4})>", body):
            for k in range(int(hi, 16) - int(lo, 16) + 1):
                table[int(lo, 16) + k] = chr(int(first, 16) + k)
    for body in re.findall(r"beginbfchar\n(.*?)\nendbfchar", cmap, re.S):
        for code, dst in re.findall(r"<([0-9A-F]{4})> <([0-9A-F]+)>", body):
            table[int(code, 16)] = bytes.fromhex(dst).decode("utf-16-be")
    return table


def test_characters_beyond_the_bmp_copy_back_whole():
    """A CJK Extension B kanji (U+20BB7, the "tsuchiyoshi" of Yoshinoya), U+2000B and an emoji among BMP text, on two pages
    of one file: each gets ONE 2-byte code of the surrogate block, the same code wherever it appears, and the ToUnicode map
    gives the character back whole - where UTF-16 code units as codes would map to two lone surrogates."""
    quad = [[100, 100], [700, 100], [700, 160], [100, 160]]

    def doc(text):
        return DocumentAnalyzerSchema(paragraphs=[ParagraphSchema(box=[50, 50, 900, 400], contents="", direction="horizontal", order=0, role=None)],
                                      tables=[], figures=[],
                                      words=[WordPrediction(points=quad, content=text, direction="horizontal", rec_score=1.0, det_score=1.0)])

    texts = ["\U00020BB7野家で\U0002000B", "丼\U0001F35A と \U00020BB7"]
    image = Image.fromarray(np.full((500, 1000, 3), 255, np.uint8))
    pages = _pages(searchable_pdf_bytes([image, image], [doc(t) for t in texts], "high"))
    font = pages[0]["font"]
    assert pages[1]["font"] == font
    cmap = pages[0]["objs"][_ref(font, "ToUnicode")][1].decode("ascii")
    table = _to_unicode_map(cmap)
    assert cmap.count("beginbfchar") == 1 and "3 beginbfchar" in cmap
    assert table[0xD800] == "\U00020BB7" and table[0xD801] == "\U0002000B" and table[0xD802] == "\U0001F35A"
    assert 0xD803 not in table and table[0x91CE] == "野"
    for page, text in zip(pages, texts):
        hexes = re.findall(r"Tm <([0-9A-F]*)> Tj", page["content"])
        assert len(hexes) == 1 and len(hexes[0]) == 4 * len(text)  # one 2-byte code per CHARACTER
        codes = [int(hexes[0][i : i + 4], 16) for i in range(0, len(hexes[0]), 4)]
        assert "".join(table[c] for c in codes) == text
    # the width model counts such a character as wide (one em), as the /DW of the font does
    assert string_width("\U00020BB7", 10) == 10


def test_a_lone_surrogate_in_the_input_becomes_the_replacement_character():
    from yomitoku_amd.utils.searchable_pdf import _hex_codes

    sup = {}
    assert _hex_codes("a\ud800b", sup) == "0061FFFD0062" and sup == {}

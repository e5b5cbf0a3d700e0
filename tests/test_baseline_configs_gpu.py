"""Parity at the configurations BASELINE.json names ("Parity at the configs BASELINE names"):

  configs[0]  OCR with the --lite configs (cli/main.py:505-520; demo/simple_ocr.py) on the reference's OWN sample images -
              demo/sample_text.jpg (91 x 38: the INTER_AREA UP-scale branch of the detector pre-processing, a one-line
              page) and tests/data/test.jpg (596 x 842) - kept as tests/golden/{sample_text,test_page}.jpg: product vs the
              oracle chain, detector input and probability map within tolerance, every discrete stage on the same upstream
              tensor, words (points, strings, directions) equal;
  configs[1]  DBNet, batch 8 of 1600 x 1200 pages (3 x 1600 x 1184 each): one page against the oracle, every page against
              its own batch-1 map (to 1e-5: at batch 1 the grid-starved 1/32-scale layers take the split-K kernel, whose
              partial sums are added in another order than conv_igemm's k-ordered chain);
  configs[2]  TextRecognizer("parseq") - open-beta geometry, FULL depth 12, refinement on - on 256 synthetic lines through
              the fixed batch_size = 128 branch of _make_mini_batch (text_recognizer.py:191-203) against
              oracle.pipeline.recognize; and the default recogniser's geometry (parseq-large-v4_1: D = 768, 8 heads of 96,
              8 x 8 patches, depth 12) on two widths against the oracle forward;
  configs[3]  one whole DocumentAnalyzerSchema against the ORACLE chain (not against the product's own stages):
              continuous stages within tolerance, every discrete stage fed the same upstream tensor on both sides,
              and the final schema equal to the (reference-pinned) aggregation of the oracle-side stage results - on a
              1000 x 1400 page and on BASELINE's 1600 x 1200;
  at the shapes the bench runs: configs[2] on all 2048 lines (16 mini-batches, two grouped forwards), and a wave-sized
              grouped forward (~650 lines, 30 mini-batches, rows-per-block of the fused greedy step on auto).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_dbnet_batch8_full_size(dev):
    from oracle.dbnet import dbnet_forward
    from oracle.preprocess import detector_preprocess
    from yomitoku_amd import imaging
    from yomitoku_amd.nets import DBNet
    from yomitoku_amd.utils.synth import dbnet_state_dict, synthetic_page

    sd = dbnet_state_dict(1234)
    net = DBNet().load_state_dict(sd).to(dev)
    imgs = [synthetic_page(30 + i, 1600, 1200) for i in range(8)]
    x = torch.cat([imaging.detector_tensor(imaging.page_to_device(im, dev), 1280, 1600) for im in imgs], 0)
    assert tuple(x.shape) == (8, 3, 1600, 1184)
    out = net(x)["binary"]
    ref = dbnet_forward(sd, detector_preprocess(imgs[5]))["binary"]
    err = (out[5:6].cpu() - ref).abs().max().item()
    assert err < 1e-3, f"page 5 of the batch: max |dP| = {err}"
    worst = 0.0
    for i in range(8):
        worst = max(worst, (net(x[i : i + 1])["binary"] - out[i : i + 1]).abs().max().item())
    assert worst < 1e-5, f"batch-8 maps differ from their batch-1 maps by {worst}"
    assert torch.equal(net(x)["binary"], out)  # the same launch shapes: bit-identical on repeat


@pytest.mark.slow(order=4)
def test_text_recognizer_open_beta_batch128_branch(dev):
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_sheet

    sheet, quads = synthetic_line_sheet(seed=5, n_lines=256)
    rec = TextRecognizer(model_name="parseq", from_pretrained=False, device="cuda:0", dynamic_width=True, batch_bucketing=True)
    assert getattr(rec._cfg.data, "width_budget", None) is None and int(rec._cfg.data.batch_size) == 128
    assert int(rec._cfg.encoder.depth) == 12 and int(rec.model.refine_iters) == 1
    sd = parseq_state_dict(1236, patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, eos_bias=6.5)
    rec.model.load_state_dict(sd)
    batches, _, dataset, order = rec.preprocess(sheet, quads)
    assert [len(b) for b in batches] == [128, 128] and order is not None
    res, _ = rec(sheet, quads)
    ocfg = make_cfg(**PRESETS["parseq"])
    contents, scores, directions = op.recognize(sd, ocfg, sheet, quads, rec.charset, dynamic_width=True, batch_bucketing=True,
                                                width_budget=None, max_batch_size=None, batch_size=128)
    assert res.contents == contents and res.directions == directions and res.points == quads
    assert np.allclose(res.scores, scores, rtol=1e-3, atol=1e-6)
    print("open-beta: distinct strings", len(set(contents)), "mean length", np.mean([len(c) for c in contents]))


@pytest.mark.parametrize("width,batch", [(800, 2), (320, 3)])
def test_parseq_large_v4_1_geometry(dev, width, batch):
    """The default recogniser (text_recognizer.py:51, cfg_text_recognizer_parseq_large_v4_1.py): per-op decoder path at
    D = 768, flash attention with head dim 96, full depth."""
    from oracle.parseq import PRESETS, make_cfg, parseq_forward
    from tests.test_parseq_gpu import _net
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    sd = parseq_state_dict(1237, patch=(8, 8), enc_dim=768, dec_dim=768, num_tokens=7121, eos_bias=6.5)
    ocfg, net = _net(dev, sd, "parseq-large-v4_1")
    assert (ocfg.enc_dim, ocfg.enc_heads, ocfg.enc_depth) == (768, 8, 12)
    x = synthetic_line_batch(40 + width, batch, width)
    ref, steps = parseq_forward(sd, ocfg, x, return_steps=True)
    out = net(x.to(dev)).cpu()
    assert net.last_ar_steps == steps and out.shape == ref.shape
    assert torch.equal(out.argmax(-1), ref.argmax(-1))
    assert (out - ref).abs().max().item() < 1e-3


@pytest.mark.slow(order=2)
def test_text_recognizer_2048_lines_every_mini_batch(dev):
    """configs[2] at its full size: ONE TextRecognizer("parseq") call on 2048 lines = 16 mini-batches of 128 through two
    grouped forwards.  The oracle chain (oracle.pipeline.recognize: crops, bucketing, batching, decode, un-permutation) runs
    over all 2048 lines; its network forward is the CPU oracle for two of the 16 mini-batches (the narrowest and a middle
    one: a 128-line full-depth forward costs tens of seconds on the host) and the product's own SINGLE-group call for the
    others - so every mini-batch of the grouped forwards is checked against its own call, two of them against the
    oracle, and strings / directions / points / scores of all 2048 lines against the chain."""
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg, parseq_forward
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_sheet

    sheet, quads = synthetic_line_sheet(seed=1, n_lines=2048)
    rec = TextRecognizer(model_name="parseq", from_pretrained=False, device="cuda:0", dynamic_width=True, batch_bucketing=True)
    sd = parseq_state_dict(1236, patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, eos_bias=6.5)
    rec.model.load_state_dict(sd)
    batches, _, _, order = rec.preprocess(sheet, quads)
    assert len(batches) == 16 and all(len(b) == 128 for b in batches) and order is not None
    res, _ = rec(sheet, quads)
    ocfg = make_cfg(**PRESETS["parseq"])
    calls = []

    def forward(x):
        k = len(calls)
        calls.append(tuple(x.shape))
        if k in (0, 8):
            return parseq_forward(sd, ocfg, x)
        return rec.model(x.to(dev)).cpu()

    contents, scores, directions = op.recognize(sd, ocfg, sheet, quads, rec.charset, dynamic_width=True, batch_bucketing=True,
                                                width_budget=None, max_batch_size=None, batch_size=128, forward=forward)
    assert len(calls) == 16 and calls[0][0] == 128
    assert res.points == quads and len(res.contents) == 2048
    assert res.contents == contents and res.directions == directions
    assert np.allclose(res.scores, scores, rtol=1e-3, atol=1e-6)
    print("2048 lines: distinct strings", len(set(contents)), "widest mini-batch", max(c[3] for c in calls), "px")


@pytest.mark.slow(order=3)
def test_wave_sized_grouped_forward(dev):
    """The shape the analyzer bench runs: ~650 lines in 30 mini-batches of one grouped forward, rows-per-block of the fused
    greedy step left on AUTO (more than 288 rows: two rows per block).  Six groups against the oracle, every group against
    its own single-group call (tokens, step counts, logits)."""
    from oracle.parseq import parseq_forward
    from tests.test_parseq_gpu import LOGIT_TOL, _groups, _net
    from yomitoku_amd.utils.synth import parseq_state_dict

    sd = parseq_state_dict(1235, eos_bias=5.5)
    ocfg, net = _net(dev, sd)
    rng = np.random.default_rng(8)
    shapes = []
    while sum(b for b, _ in shapes) < 640 or len(shapes) < 30:
        w = int(rng.choice([72, 96, 128, 160, 200, 240, 320, 400, 560, 800]))
        shapes.append((int(min(32, max(1, 8000 // w - int(rng.integers(0, 4))))), w))
    xs = _groups(300, shapes)
    rows = sum(x.shape[0] for x in xs)
    assert rows >= 600 and len(xs) >= 25
    logits, out_lens, steps = net.forward_groups([x.to(dev) for x in xs])
    lg = logits.cpu()
    assert len(set(steps)) > 3, "groups should stop at different steps"
    small = sorted(range(len(xs)), key=lambda g: xs[g].shape[0] * xs[g].shape[3])
    oracle_groups = set(small[:4] + small[len(small) // 2 : len(small) // 2 + 2])
    row = 0
    for g, (x, n, st) in enumerate(zip(xs, out_lens, steps)):
        got = lg[row : row + x.shape[0], :n]
        one = net(x.to(dev)).cpu()
        assert net.last_ar_steps == st, g
        assert torch.equal(one.argmax(-1), got.argmax(-1)) and (one - got).abs().max().item() < 1e-4, g
        if g in oracle_groups:
            ref, ref_steps = parseq_forward(sd, ocfg, x, return_steps=True)
            assert st == ref_steps and got.shape == ref.shape, g
            assert torch.equal(got.argmax(-1), ref.argmax(-1)) and (got - ref).abs().max().item() < LOGIT_TOL, g
        row += x.shape[0]
    again, _, _ = net.forward_groups([x.to(dev) for x in xs])
    assert torch.equal(again.cpu(), lg), "bit-identical on repeat"
    print("wave-sized forward:", rows, "lines,", len(xs), "groups, steps", min(steps), "..", max(steps))


@pytest.mark.slow(order=1)
@pytest.mark.parametrize("page_hw", [(1600, 1200), (1000, 1400)])
def test_whole_page_schema_vs_oracle_chain(dev, page_hw):
    from oracle import pipeline as op
    from oracle.dbnet import dbnet_forward
    from oracle.parseq import PRESETS, make_cfg
    from oracle.preprocess import detector_preprocess
    from tests.test_pipeline_gpu import _assert_same_schema
    from tests.test_rtdetr_gpu import assert_same_detections
    from yomitoku_amd import DocumentAnalyzer
    from yomitoku_amd.document_analyzer import ocr_aggregate
    from yomitoku_amd.schemas import DocumentAnalyzerSchema, LayoutAnalyzerSchema, OCRSchema, TextDetectorSchema, TextRecognizerSchema
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    img = synthetic_page_with_truth(3, *page_hw)[0]
    configs = {
        "ocr": {"text_detector": {"from_pretrained": False},
                "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                                    "batch_bucketing": True, "source_downscale": True}},
        "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}},
    }
    an = DocumentAnalyzer(configs=configs, device="cuda:0")
    sds = {"det": dbnet_state_dict(1234, out_bias=-2.0), "rec": parseq_state_dict(1235, eos_bias=6.0),
           # table seed 1243: one row and many columns clear the 0.4 threshold on this page's table crops (seeds 1241 / 1245 fire on
           # one class only, so every table would be dropped for lack of rows or columns and the cell / aggregation path idle)
           "lay": rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0), "tab": rtdetr_state_dict(1243, num_classes=3, score_bias=-1.0)}
    an.text_detector.model.load_state_dict(sds["det"])
    an.text_recognizer.model.load_state_dict(sds["rec"])
    an.layout.layout_parser.model.load_state_dict(sds["lay"])
    an.layout.table_structure_recognizer.model.load_state_dict(sds["tab"])
    got, _, _ = an(img)

    # ---- detector: continuous stage vs the oracle, discrete stage on the same upstream tensor
    det = an.text_detector
    prob = det.model(det.preprocess(img))["binary"].cpu()
    ref_prob = dbnet_forward(sds["det"], detector_preprocess(img))["binary"]
    assert (prob - ref_prob).abs().max().item() < 1e-3
    _, quads, det_scores = op.detect(sds["det"], img, prob=prob)
    assert len(quads) >= 5
    # ---- recogniser: the oracle chain on those quads
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    contents, rec_scores, directions = op.recognize(sds["rec"], ocfg, img, quads, an.text_recognizer.charset, dynamic_width=True,
                                                    batch_bucketing=True, width_budget=8000, max_batch_size=64, batch_size=10,
                                                    source_downscale=True)
    # ---- layout: logits / boxes vs the oracle forward; the box logic on the product's tensor (same upstream policy)
    lp, ts = an.layout.layout_parser, an.layout.table_structure_recognizer
    preds = lp.model(lp.preprocess(img))
    ref_preds, _ = op.layout(sds["lay"], img)
    assert_same_detections(preds["pred_logits"].cpu().numpy(), preds["pred_boxes"].cpu().numpy(),
                           ref_preds["pred_logits"].numpy(), ref_preds["pred_boxes"].numpy())
    lay = lp.postprocess(preds, img.shape[:2])
    # ---- from here on the ORACLE's own host logic (oracle/hostlogic.py, pinned against the reference's functions): box filters,
    # cell grids, aggregation and reading order owe nothing to yomitoku_amd - until round 5 `want` was built by an.aggregate
    from oracle import hostlogic as hl

    h, w = img.shape[:2]
    groups = hl.layout_elements(op.rtdetr_post(preds["pred_logits"].cpu(), preds["pred_boxes"].cpu(), (w, h), 0.5, 6)[0])
    for key, mine in (("tables", lay.tables), ("paragraphs", lay.paragraphs), ("figures", lay.figures)):  # product's filters == oracle's
        _assert_same_schema(groups[key], [e.model_dump() for e in mine], score_rtol=1e-6)
    tables = []
    if lay.tables:
        batch, metas = ts.preprocess(img, [t.box for t in lay.tables])
        tp = ts.model(batch)
        for i, (t, meta) in enumerate(zip(lay.tables, metas)):
            if i < 4:  # each table costs an oracle forward on the CPU: the first four against the oracle, all through the product
                (rp, _), = op.tables(sds["tab"], img, [t.box])
                assert_same_detections(tp["pred_logits"][i : i + 1].cpu().numpy(), tp["pred_boxes"][i : i + 1].cpu().numpy(),
                                       rp["pred_logits"].numpy(), rp["pred_boxes"].numpy())
            th, tw = meta["size"]
            det = op.rtdetr_post(tp["pred_logits"][i : i + 1].cpu(), tp["pred_boxes"][i : i + 1].cpu(), (tw, th), 0.4, 3)[0]
            table = hl.table_structure(det, (th, tw), meta["offset"])
            _assert_same_schema(table, ts.postprocess({"pred_logits": tp["pred_logits"][i : i + 1], "pred_boxes": tp["pred_boxes"][i : i + 1]}, meta).model_dump(),
                                score_rtol=1e-6)
            if table["n_row"] > 0 and table["n_col"] > 0:
                tables.append(table)
    # ---- the page record the reference's aggregation builds from the ORACLE-side stage results
    words = [{"points": [[int(x), int(y)] for x, y in q], "content": c, "direction": d, "rec_score": float(rs), "det_score": float(ds)}
             for q, ds, c, rs, d in zip(quads, det_scores, contents, rec_scores, directions)]
    want = DocumentAnalyzerSchema(**hl.aggregate(words, {"paragraphs": groups["paragraphs"], "tables": tables, "figures": groups["figures"]}))
    try:
        _assert_same_schema(want.model_dump(), got.model_dump(), score_rtol=1e-3)
    except AssertionError:  # say which elements differ before failing (the first differing leaf alone rarely tells)
        for name, d in (("oracle side", want), ("product", got)):
            print(name, "paragraphs", [(p.box, p.order, p.direction, p.contents[:12]) for p in d.paragraphs])
            print(name, "tables", [(t.box, t.n_row, t.n_col, t.order, len(t.cells)) for t in d.tables])
            print(name, "figures", [(f.box, f.order, len(f.paragraphs)) for f in d.figures])
            print(name, "words", len(d.words), [(w.content[:8], w.direction, round(w.rec_score, 4)) for w in d.words][:40])
        raise
    assert len(got.words) == len(quads) and sum(len(w.content) for w in got.words) > 0
    if page_hw == (1000, 1400):  # the layout net finds table boxes on this page and at least one keeps rows AND columns
        assert len(lay.tables) >= 1 and len(got.tables) >= 1 and sum(len(t.cells) for t in got.tables) >= 1
    print("whole page: words", len(got.words), "paragraphs", len(got.paragraphs), "tables", len(got.tables), "figures", len(got.figures))


def _sample(name):
    """(BGR page through the product's loader, the same file decoded independently) of tests/golden/<name>."""
    import os

    from PIL import Image

    from yomitoku_amd.data.functions import load_image

    path = os.path.join(os.path.dirname(__file__), "golden", name)
    (img,) = load_image(path)
    ref_img = np.ascontiguousarray(np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])
    assert np.array_equal(img, ref_img)
    return img, ref_img


LITE_OCR = {"text_detector": {"from_pretrained": False},
            "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                                "batch_bucketing": True, "source_downscale": True}}


def test_ocr_lite_on_the_reference_test_page(dev):
    """BASELINE.json configs[0] on tests/data/test.jpg (596 x 842): OCR(--lite configs) against the oracle chain.  The
    reference decodes with Pillow and hands BGR to OCR (data/functions.py:57-77); the oracle side decodes the file on its
    own, the product side goes through yomitoku_amd.data.load_image."""
    from oracle import pipeline as op
    from oracle.dbnet import dbnet_forward
    from oracle.parseq import PRESETS, make_cfg
    from oracle.preprocess import detector_preprocess
    from tests.test_pipeline_gpu import _assert_same_schema
    from yomitoku_amd import OCR
    from yomitoku_amd.document_analyzer import ocr_aggregate
    from yomitoku_amd.schemas import OCRSchema, TextDetectorSchema, TextRecognizerSchema
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict

    img, ref_img = _sample("test_page.jpg")
    ocr = OCR(configs=LITE_OCR, device="cuda:0")
    # seed 8 / bias -1.5: the seeded detector finds ~90 boxes on this page, every one a valid crop (most seeds find none)
    sds = {"det": dbnet_state_dict(8, out_bias=-1.5), "rec": parseq_state_dict(1235, eos_bias=6.0)}
    ocr.detector.model.load_state_dict(sds["det"])
    ocr.recognizer.model.load_state_dict(sds["rec"])
    got, _ = ocr(img)

    det = ocr.detector
    t = det.preprocess(img)
    ref_t = detector_preprocess(ref_img)
    assert tuple(t.shape) == tuple(ref_t.shape) == (1, 3, 1600, 1120)
    assert (t.cpu() - ref_t).abs().max().item() < 2e-5
    prob = det.model(t)["binary"].cpu()
    assert (prob - dbnet_forward(sds["det"], ref_t)["binary"]).abs().max().item() < 1e-3
    _, quads, det_scores = op.detect(sds["det"], ref_img, prob=prob)
    assert len(quads) >= 40, len(quads)
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    contents, rec_scores, directions = op.recognize(sds["rec"], ocfg, ref_img, quads, ocr.recognizer.charset, dynamic_width=True,
                                                    batch_bucketing=True, width_budget=8000, max_batch_size=64, batch_size=10,
                                                    source_downscale=True)
    want = OCRSchema(words=ocr_aggregate(TextDetectorSchema(points=quads, scores=det_scores),
                                         TextRecognizerSchema(contents=contents, scores=rec_scores, points=quads, directions=directions)))
    _assert_same_schema(want.model_dump(), got.model_dump(), score_rtol=1e-3)
    assert sum(len(w.content) for w in got.words) > 0
    print("test_page.jpg: words", len(got.words), [w.content for w in got.words[:3]])


def test_ocr_lite_on_the_reference_sample_text(dev):
    """BASELINE.json configs[0] proper: demo/sample_text.jpg, 91 x 38 - the detector pre-processing scales it UP by 16.8
    (resize_shortest_edge, data/functions.py:196-227: the INTER_AREA branch no other test image reaches) and the page is one
    text line.  A seeded detector's blobs shrink to points when its 640 x 1568 map is scaled back to 38 x 91 (the reference
    hands OpenCV an empty dsize for those and raises; the product reports them as empty strings), so the
    chain is checked in its two halves: detector stage on the real image (input tensor, map, boxes on the same map), and the
    recogniser on the real image with the line's own quad - what a trained detector returns for a one-line page."""
    from oracle import pipeline as op
    from oracle.dbnet import dbnet_forward
    from oracle.parseq import PRESETS, make_cfg
    from oracle.preprocess import detector_preprocess
    from yomitoku_amd import OCR
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict

    img, ref_img = _sample("sample_text.jpg")
    assert img.shape == (38, 91, 3)
    ocr = OCR(configs=LITE_OCR, device="cuda:0")
    sds = {"det": dbnet_state_dict(9, out_bias=-0.5), "rec": parseq_state_dict(1235, eos_bias=6.0)}
    ocr.detector.model.load_state_dict(sds["det"])
    ocr.recognizer.model.load_state_dict(sds["rec"])
    det, rec = ocr.detector, ocr.recognizer
    t = det.preprocess(img)
    ref_t = detector_preprocess(ref_img)
    assert tuple(t.shape) == tuple(ref_t.shape) == (1, 3, 640, 1568)
    assert (t.cpu() - ref_t).abs().max().item() < 2e-5
    prob = det.model(t)["binary"]
    assert (prob.cpu() - dbnet_forward(sds["det"], ref_t)["binary"]).abs().max().item() < 1e-3
    res, _ = det(img)
    _, quads, det_scores = op.detect(sds["det"], ref_img, prob=prob.cpu())
    assert len(quads) >= 1 and res.points == quads
    assert np.allclose(res.scores, det_scores, atol=1e-9)
    words, _ = ocr(img)  # degenerate boxes must come back as aligned placeholders, not as an exception
    assert [w.points for w in words.words] == quads

    line = [[[0, 0], [90, 0], [90, 37], [0, 37]], [[3, 4], [60, 4], [60, 33], [3, 33]], [[30, 2], [88, 6], [86, 36], [28, 30]]]
    got, _ = rec(img, line)
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    contents, scores, directions = op.recognize(sds["rec"], ocfg, ref_img, line, rec.charset, dynamic_width=True, batch_bucketing=True,
                                                width_budget=8000, max_batch_size=64, batch_size=10, source_downscale=True)
    assert got.contents == contents and got.directions == directions and got.points == line
    assert np.allclose(got.scores, scores, rtol=1e-3, atol=1e-6)
    assert any(len(c) > 0 for c in contents)
    print("sample_text.jpg: boxes", len(quads), "line ->", contents)

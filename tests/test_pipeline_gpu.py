"""Module-level parity on a synthetic page, stage by stage, through the product classes.

Tolerance policy (SURVEY §7 "bit-exact indices vs 1e-3 maps"): a continuous stage is compared with
the oracle within its tolerance; a discrete stage (thresholds, contours, top-k, arg-max, integer
boxes) is compared bit-exactly with the oracle applied to THE SAME upstream tensor the product
produced, so that a 1e-6 wobble of a probability sitting on a threshold cannot masquerade as a
post-processing bug (or hide one)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def page():
    from yomitoku_amd.utils.synth import synthetic_page_with_truth

    return synthetic_page_with_truth(3, 1000, 1400)[:3]


def test_text_detector_stages(dev, page):
    from oracle import pipeline as op
    from oracle.dbnet import dbnet_forward
    from oracle.preprocess import detector_preprocess
    from yomitoku_amd.text_detector import TextDetector
    from yomitoku_amd.utils.synth import dbnet_state_dict

    img = page[0]
    det = TextDetector(from_pretrained=False, device="cuda:0")
    sd = dbnet_state_dict(1234, out_bias=-2.0)
    det.model.load_state_dict(sd)
    t = det.preprocess(img)
    ref_t = detector_preprocess(img)
    assert (t.cpu() - ref_t).abs().max().item() < 2e-5
    prob = det.model(t)["binary"]
    assert (prob.cpu() - dbnet_forward(sd, ref_t)["binary"]).abs().max().item() < 1e-3
    res, _ = det(img)
    _, quads, scores = op.detect(sd, img, prob=prob.cpu())
    assert len(quads) >= 5
    assert res.points == quads
    assert np.allclose(res.scores, scores, atol=1e-9)


def test_text_recognizer_lite_config(dev, page):
    """--lite recogniser (parseq-tiny-dynw-v4, dynamic_width + batch_bucketing): crops, batching,
    forward, decode, un-permutation."""
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict

    img, quads, _ = page
    quads = quads[:26]
    rec = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True,
                         batch_bucketing=True)
    sd = parseq_state_dict(1235, eos_bias=6.0)
    rec.model.load_state_dict(sd)
    res, _ = rec(img, quads)
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    contents, scores, directions = op.recognize(sd, ocfg, img, quads, rec.charset, dynamic_width=True,
                                                batch_bucketing=True, width_budget=8000, max_batch_size=64, batch_size=10)
    assert res.contents == contents
    assert res.directions == directions
    assert np.allclose(res.scores, scores, rtol=1e-3, atol=1e-6)
    assert res.points == quads


def test_text_recognizer_source_downscale_and_orientation_fallback(dev, page):
    """The two optional recogniser paths (`--lite` sets source_downscale; text_recognizer.py:319-350 retries
    low-score lines turned by 180 degrees): same contents / scores as the oracle chain."""
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict

    img, quads, _ = page
    quads = quads[:12] + [[[40, 60], [560, 60], [560, 200], [40, 200]], [[600, 300], [1300, 300], [1300, 600], [600, 600]],
                          [[100, 650], [420, 650], [420, 730], [100, 730]]]  # short sides 140 / 300 / 80: levels 2 / 3 / 1
    sd = parseq_state_dict(1235, eos_bias=6.0)
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    kw = dict(dynamic_width=True, batch_bucketing=True, width_budget=8000, max_batch_size=64, batch_size=10)
    plain = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True,
                           batch_bucketing=True, source_downscale=True)
    plain.model.load_state_dict(sd)
    base, _ = plain(img, quads)
    contents, scores, directions = op.recognize(sd, ocfg, img, quads, plain.charset, source_downscale=True, **kw)
    assert base.contents == contents and base.directions == directions
    assert np.allclose(base.scores, scores, rtol=1e-3, atol=1e-6)
    thresh = float(np.sort(base.scores)[len(base.scores) // 2]) * 1.0001  # about half of the lines get the retry
    rec = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True,
                         batch_bucketing=True, source_downscale=True, rec_orientation_fallback=True,
                         rec_orientation_fallback_thresh=thresh)
    rec.model.load_state_dict(sd)
    res, _ = rec(img, quads)
    contents, scores, directions = op.recognize(sd, ocfg, img, quads, rec.charset, source_downscale=True,
                                                orientation_fallback=True, fallback_thresh=thresh, **kw)
    assert res.contents == contents and res.directions == directions
    assert np.allclose(res.scores, scores, rtol=1e-3, atol=1e-6)
    assert res.points == quads
    replaced = sum(a != b for a, b in zip(res.scores, base.scores))
    print("orientation fallback replaced", replaced, "of", len(quads))


def test_text_recognizer_pages_equal_single_calls(dev, page):
    """recognize_pages: the mini-batches of several pages share PARSeq forwards; every page's strings, directions and
    scores are those of its own __call__ (mini-batches never mix pages), also when a forward is split by the line cap."""
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_page_with_truth

    img, quads, _ = page
    img2, quads2, _, _ = synthetic_page_with_truth(7)
    sd = parseq_state_dict(1235, eos_bias=6.0)
    rec = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True,
                         batch_bucketing=True, num_parallel_batches=3)
    rec.model.load_state_dict(sd)
    rec._cfg.data.max_batch_size = 8  # several mini-batches per page
    singles = [rec(i, q)[0] for i, q in ((img, quads), (img2, quads2), (img, quads[:1]))]
    for cap in (1024, 20):
        rec.MAX_LINES_PER_FORWARD = cap
        multi = rec.recognize_pages([img, img2, img], [quads, quads2, quads[:1]])
        for a, b in zip(singles, multi):
            assert a.contents == b.contents and a.directions == b.directions and a.points == b.points
            assert np.allclose(a.scores, b.scores, rtol=1e-4, atol=1e-7)
    assert rec.recognize_pages([], []) == []


def test_layout_and_table_stages(dev, page):
    from oracle import pipeline as op
    from oracle.rtdetr import rtdetr_forward
    from tests.test_rtdetr_gpu import assert_same_detections
    from yomitoku_amd.layout_parser import LayoutParser
    from yomitoku_amd.table_structure_recognizer import TableStructureRecognizer
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    img, _, tables = page
    lp = LayoutParser(from_pretrained=False, device="cuda:0")
    sd = rtdetr_state_dict(1240, num_classes=6, score_bias=-1.5)
    lp.model.load_state_dict(sd)
    x = lp.preprocess(img)
    preds = lp.model(x)
    ref_preds, _ = op.layout(sd, img)
    assert_same_detections(preds["pred_logits"].cpu().numpy(), preds["pred_boxes"].cpu().numpy(),
                           ref_preds["pred_logits"].numpy(), ref_preds["pred_boxes"].numpy())
    h, w = img.shape[:2]
    mine = lp.postprocessor(preds, (w, h), lp.thresh_score)[0]
    ref = op.rtdetr_post(preds["pred_logits"].cpu(), preds["pred_boxes"].cpu(), (w, h), lp.thresh_score, 6)[0]
    assert len(ref["scores"]) >= 3
    assert np.array_equal(mine["labels"], ref["labels"])
    assert np.array_equal(mine["boxes"].astype(int), ref["boxes"].astype(int))
    assert np.allclose(mine["scores"], ref["scores"], atol=1e-6)
    results, _ = lp(img)
    assert len(results.paragraphs) + len(results.tables) + len(results.figures) > 0

    ts = TableStructureRecognizer(from_pretrained=False, device="cuda:0")
    sd_t = rtdetr_state_dict(1241, num_classes=3, score_bias=-1.0)
    ts.model.load_state_dict(sd_t)
    batch, metas = ts.preprocess(img, tables)
    preds_t = ts.model(batch)
    for i, (box, meta) in enumerate(zip(tables, metas)):
        (rp, _), = op.tables(sd_t, img, [box])
        assert_same_detections(preds_t["pred_logits"][i : i + 1].cpu().numpy(), preds_t["pred_boxes"][i : i + 1].cpu().numpy(),
                               rp["pred_logits"].numpy(), rp["pred_boxes"].numpy())
    out, _ = ts(img, tables)
    for t in out:
        assert t.n_row > 0 and t.n_col > 0 and len(t.cells) > 0


def test_document_analyzer_end_to_end(dev, page):
    """The orchestrator returns exactly what its stages return when run one by one (two worker threads
    on separate HIP streams must not perturb results), and the schema is well formed."""
    from yomitoku_amd import DocumentAnalyzer
    from yomitoku_amd.document_analyzer import ocr_aggregate
    from yomitoku_amd.schemas import DocumentAnalyzerSchema, OCRSchema

    img = page[0]
    configs = {
        "ocr": {
            "text_detector": {"from_pretrained": False},
            "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                                "batch_bucketing": True},
        },
        "layout_analyzer": {"layout_parser": {"from_pretrained": False},
                            "table_structure_recognizer": {"from_pretrained": False}},
    }
    an = DocumentAnalyzer(configs=configs, device="cuda:0")
    from yomitoku_amd.utils.synth import dbnet_state_dict

    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    an.text_detector.model.load_state_dict(dbnet_state_dict(1234, out_bias=-2.0))
    an.layout.layout_parser.model.load_state_dict(rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0))
    an.layout.table_structure_recognizer.model.load_state_dict(rtdetr_state_dict(1241, num_classes=3, score_bias=-1.0))
    results, ocr_vis, layout_vis = an(img)
    assert isinstance(results, DocumentAnalyzerSchema) and ocr_vis is None
    det, _ = an.text_detector(img)
    rec, _ = an.text_recognizer(img, det.points)
    lay, _ = an.layout(img)
    an.img = img
    expect = DocumentAnalyzerSchema(**an.aggregate(OCRSchema(words=ocr_aggregate(det, rec)), lay))
    assert results.model_dump() == expect.model_dump()
    again, _, _ = an(img)
    assert again.model_dump() == results.model_dump()
    orders = sorted([p.order for p in results.paragraphs] + [t.order for t in results.tables] + [f.order for f in results.figures])
    assert orders == list(range(len(orders)))


def _assert_same_schema(a, b, score_rtol=1e-4, box_tol=0, path="$"):
    """Two model_dump() trees: same structure, strings and integers; floats within score_rtol; integer box / point
    coordinates within box_tol pixels.  A failure names the path of the leaf."""
    assert type(a) is type(b), (path, a, b)
    if isinstance(a, dict):
        assert a.keys() == b.keys(), path
        for k in a:
            if k in ("box", "points") and box_tol:
                assert np.abs(np.asarray(a[k]) - np.asarray(b[k])).max() <= box_tol, (f"{path}.{k}", a[k], b[k])
            else:
                _assert_same_schema(a[k], b[k], score_rtol, box_tol, f"{path}.{k}")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), (path, len(a), len(b))
        for i, (x, y) in enumerate(zip(a, b)):
            _assert_same_schema(x, y, score_rtol, box_tol, f"{path}[{i}]")
    elif isinstance(a, float):
        assert abs(a - b) <= score_rtol * max(abs(a), abs(b)) + 1e-9, (path, a, b)
    else:
        assert a == b, (path, a, b)


def test_analyze_pages_equals_per_page_calls(dev, page):
    """DocumentAnalyzer.analyze_pages shares device batches across the pages of a wave (DBNet / RT-DETR forwards over
    several pages, one grouped PARSeq forward).  Every page's DocumentAnalyzerSchema must be the one `__call__` returns
    for that page alone: same words, strings, cells, paragraphs and reading order; scores to 1e-4 (the implicit-GEMM
    kernel picks its tile / split-K shape from the launch's M, so logits may differ in their last bits)."""
    from yomitoku_amd import DocumentAnalyzer
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    configs = {
        "ocr": {
            "text_detector": {"from_pretrained": False},
            "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                                "batch_bucketing": True, "source_downscale": True},
        },
        "layout_analyzer": {"layout_parser": {"from_pretrained": False},
                            "table_structure_recognizer": {"from_pretrained": False}},
    }
    an = DocumentAnalyzer(configs=configs, device="cuda:0")
    an.text_detector.model.load_state_dict(dbnet_state_dict(1234, out_bias=-2.0))
    an.text_recognizer.model.load_state_dict(parseq_state_dict(1235, eos_bias=6.0))
    an.layout.layout_parser.model.load_state_dict(rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0))
    an.layout.table_structure_recognizer.model.load_state_dict(rtdetr_state_dict(1241, num_classes=3, score_bias=-1.0))
    imgs = [page[0], synthetic_page_with_truth(5, 1000, 1400)[0], synthetic_page_with_truth(6, 1400, 1000)[0], page[0]]
    singles = [an(img)[0].model_dump() for img in imgs]
    assert sum(len(s["words"]) for s in singles) > 0
    for wave in (4, 3):
        multi = an.analyze_pages(imgs, wave=wave)
        assert len(multi) == len(imgs)
        for one, (many, ocr_vis, lay_vis) in zip(singles, multi):
            assert ocr_vis is None and lay_vis is None
            _assert_same_schema(one, many.model_dump())
    assert an.analyze_pages([]) == []


def test_nested_configs_reach_the_modules(dev, tmp_path):
    """tests/test_document_analyzer.py::test_initialize and tests/test_ocr.py::test_ocr of the reference: per-module YAML
    overrides given through the nested `configs` dict land in the right module (on the cuda device instead of "cpu")."""
    import torch

    from yomitoku_amd.document_analyzer import OCR, DocumentAnalyzer

    files = {}
    for name, text in (("text_detector", "post_process:\n  thresh: 0.4\n"), ("text_recognizer", "refine_iters: 0\n"),
                       ("layout_parser", "thresh_score: 0.8\n"), ("table_structure_recognizer", "thresh_score: 0.8\n")):
        files[name] = tmp_path / f"{name}.yaml"
        files[name].write_text(text)
    lite = {"from_pretrained": False}
    configs = {
        "ocr": {"text_detector": {"path_cfg": str(files["text_detector"]), **lite},
                "text_recognizer": {"path_cfg": str(files["text_recognizer"]), "model_name": "parseq-tiny-dynw-v4", **lite}},
        "layout_analyzer": {"layout_parser": {"path_cfg": str(files["layout_parser"]), **lite},
                            "table_structure_recognizer": {"path_cfg": str(files["table_structure_recognizer"]), **lite}},
    }
    an = DocumentAnalyzer(configs=configs, device="cuda:0", visualize=False)
    for module in (an.text_detector, an.text_recognizer, an.layout.layout_parser, an.layout.table_structure_recognizer):
        assert torch.device(module.device) == torch.device("cuda:0") and module.visualize is False
    assert an.text_detector.post_processor.thresh == 0.4
    assert an.text_recognizer.model.refine_iters == 0
    assert an.layout.layout_parser.thresh_score == 0.8
    assert an.layout.table_structure_recognizer.thresh_score == 0.8
    ocr = OCR(configs=configs["ocr"], device="cuda:0")
    assert ocr.detector.post_processor.thresh == 0.4 and ocr.recognizer.model.refine_iters == 0


def test_degenerate_quad_gets_a_placeholder_and_the_outputs_stay_aligned(dev, page, caplog):
    """A quad that passes validate_quads but whose first edge is shorter than one pixel cannot be cropped (the reference
    hands OpenCV an empty dsize).  It is reported as an empty string with score 0 - loudly - so that contents / scores /
    directions keep lining up with `points` and the page's other lines are what they are without it."""
    import logging

    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict

    img, quads, _ = page
    quads = [list(map(list, q)) for q in quads[:9]]
    rec = TextRecognizer(model_name="parseq-tiny-dynw-v4", from_pretrained=False, device="cuda:0", dynamic_width=True,
                         batch_bucketing=True)
    rec.model.load_state_dict(parseq_state_dict(1235, eos_bias=6.0))
    rec.batch_bucketing = False  # the call with a dropped quad below runs un-bucketed too (text_recognizer.py:139)
    clean, _ = rec(img, quads)
    rec.batch_bucketing = True
    bad = [[50, 60], [50, 60], [50, 200], [50, 200]]  # zero-length first edge, tall: "vertical"
    with caplog.at_level(logging.WARNING, logger="yomitoku_amd.text_recognizer"):
        mixed, _ = rec(img, quads[:4] + [bad] + quads[4:])
    assert "shorter" in caplog.text
    assert len(mixed.contents) == len(mixed.scores) == len(mixed.directions) == len(mixed.points) == 10
    assert (mixed.contents[4], mixed.scores[4], mixed.directions[4]) == ("", 0.0, "vertical")
    # the other lines went through the same mini-batches in the same order: same strings
    assert mixed.contents[:4] + mixed.contents[5:] == clean.contents
    many = rec.recognize_pages([img], [quads[:4] + [bad] + quads[4:]])[0]
    assert many.contents == mixed.contents and many.directions == mixed.directions

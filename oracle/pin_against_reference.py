"""ORACLE tooling: pin the oracle restatements against the reference's own code (run in the
build container where /root/reference exists) and write the golden vectors under tests/golden/.

    python -m oracle.pin_against_reference [dbnet|parseq|rtdetr|all]

Every golden file records the seed of the synthetic checkpoint (yomitoku_amd/utils/synth.py), the
input, and the output of the REFERENCE implementation; tests compare both the oracle (CPU) and the
HIP path (GPU box) against it.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from ._refstubs import AttrDict, ref_import

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def pin_dbnet():
    from yomitoku_amd.utils.synth import dbnet_state_dict

    from .dbnet import dbnet_forward

    mod = ref_import("yomitoku.models.dbnet_plus")
    cfg = AttrDict(
        backbone={"name": "resnet50", "dilation": True},
        decoder={"in_channels": [256, 512, 1024, 2048], "hidden_dim": 256, "adaptive": True, "serial": True,
                 "smooth": False, "k": 50},
    )
    seed = 1234
    sd = dbnet_state_dict(seed)
    model = mod.DBNet(cfg)
    missing = model.load_state_dict(sd, strict=True)  # key names + shapes must match the reference exactly
    model.eval()
    x = torch.randn(1, 3, 64, 96, generator=torch.Generator().manual_seed(64))
    with torch.inference_mode():
        ref = model(x)["binary"]
    ours = dbnet_forward(sd, x)["binary"]
    err = (ref - ours).abs().max().item()
    print(f"[dbnet] reference vs oracle: max abs diff {err:.3e} ({missing})")
    assert err < 1e-6, err
    np.savez_compressed(os.path.join(GOLDEN, "dbnet_ref_64x96.npz"), seed=seed, x=x.numpy(), prob=ref.numpy())


def pin_parseq():
    from types import SimpleNamespace

    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_batch

    from .parseq import PRESETS, make_cfg, parseq_forward

    mod = ref_import("yomitoku.models.parseq")
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    rcfg = AttrDict(
        max_label_length=100, decode_ar=1, refine_iters=1, num_tokens=ocfg.num_tokens,
        data={"img_size": [32, 800]},
        encoder={"patch_size": [4, 8], "num_heads": 6, "embed_dim": 192, "mlp_ratio": 4, "depth": 12},
        decoder={"embed_dim": 192, "num_heads": 6, "mlp_ratio": 4, "depth": 1},
    )
    cases = [
        # (file tag, checkpoint kwargs, batch, width)
        ("eos", dict(seed=1235, eos_bias=4.5), 3, 96),
        ("rep", dict(seed=1236, eos_bias=3.0, favour_token=17, favour_bias=12.0), 3, 72),
    ]
    for tag, kw, bs, width in cases:
        sd = parseq_state_dict(**kw)
        model = mod.PARSeq(rcfg)
        res = model.load_state_dict(sd, strict=True)
        model.eval()
        model.tokenizer = SimpleNamespace(eos_id=0, bos_id=ocfg.num_tokens - 2, pad_id=ocfg.num_tokens - 1)
        x = synthetic_line_batch(11, bs, width)
        with torch.inference_mode():
            ref = model(x)
        ours, steps = parseq_forward(sd, ocfg, x, return_steps=True)
        err = (ref - ours).abs().max().item()
        ids = ref.argmax(-1)
        lens = [int((row == 0).nonzero()[0]) if (row == 0).any() else len(row) for row in ids]
        print(f"[parseq/{tag}] reference vs oracle: max abs diff {err:.3e}, AR steps {steps}, lengths {lens} ({res})")
        assert err < 1e-5, err
        # logits are large (B x 101 x 7119): keep the decisive slices - arg-max ids, max logit, a strided sample
        np.savez_compressed(
            os.path.join(GOLDEN, f"parseq_ref_{tag}.npz"), ckpt=np.array(repr(kw)), x=x.numpy(), steps=steps,
            ids=ids.numpy().astype(np.int32), top=ref.max(-1).values.numpy(), sample=ref[:, :, ::97].numpy(),
        )


def pin_rtdetr():
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    from .rtdetr import rtdetr_forward

    mod = ref_import("yomitoku.models.rtdetr")
    import omegaconf  # the stub: RTDETRTransformerv2 wants num_points as a ListConfig

    for tag, nc, seed, size, nq in (("layout", 6, 1240, 640, 300), ("table", 3, 1241, 640, 300), ("cell", 6, 1243, 960, 1500)):
        cfg = AttrDict(
            PResNet={"depth": 50, "variant": "d", "freeze_at": 0, "return_idx": [1, 2, 3], "num_stages": 4,
                     "freeze_norm": True},
            HybridEncoder={"in_channels": [512, 1024, 2048], "feat_strides": [8, 16, 32], "hidden_dim": 256,
                           "use_encoder_idx": [2], "num_encoder_layers": 1, "nhead": 8, "dim_feedforward": 1024,
                           "dropout": 0.0, "enc_act": "gelu", "expansion": 1.0, "depth_mult": 1, "act": "silu"},
            RTDETRTransformerv2={"num_classes": nc, "feat_channels": [256, 256, 256], "feat_strides": [8, 16, 32],
                                 "hidden_dim": 256, "num_levels": 3, "num_layers": 6, "num_queries": nq,
                                 "num_denoising": 100, "label_noise_ratio": 0.5, "box_noise_scale": 1.0,
                                 "eval_spatial_size": [size, size], "eval_idx": -1,
                                 "num_points": omegaconf.ListConfig([4, 4, 4]), "cross_attn_method": "default",
                                 "query_select_method": "default"},
        )
        sd = rtdetr_state_dict(seed, num_classes=nc, eval_size=(size, size), enc_score_gain=1.0 if size == 640 else 12.0)
        model = mod.RTDETRv2(cfg)
        res = model.load_state_dict(sd, strict=True)
        model.eval()
        x = torch.rand(1, 3, size, size, generator=torch.Generator().manual_seed(seed))
        with torch.inference_mode():
            ref = model(x)
        ours = rtdetr_forward(sd, x, num_queries=nq)
        e1 = (ref["pred_logits"] - ours["pred_logits"]).abs().max().item()
        e2 = (ref["pred_boxes"] - ours["pred_boxes"]).abs().max().item()
        sc = ref["pred_logits"].sigmoid()
        print(f"[rtdetr/{tag}] reference vs oracle: logits {e1:.3e} boxes {e2:.3e}; scores>0.5: {(sc > 0.5).sum().item()} ({res})")
        if e1 >= 1e-5 or e2 >= 1e-6:
            # 1500 of 18900 tokens: two query candidates whose encoder scores differ by fp32 summation noise can swap
            # ranks between two implementations (torch.topk's order under near-ties is not defined); rows are then
            # permuted, not different - match them one to one inside a +-2 rank band, as the GPU test does
            from tests.test_rtdetr_gpu import assert_same_detections

            assert_same_detections(ours["pred_logits"].numpy(), ours["pred_boxes"].numpy(), ref["pred_logits"].numpy(),
                                   ref["pred_boxes"].numpy(), tol_logit=2e-5, tol_box=2e-6)
            print(f"[rtdetr/{tag}]   rows equal up to near-tie rank swaps (2e-5 / 2e-6 after one-to-one matching)")
        np.savez_compressed(os.path.join(GOLDEN, f"rtdetr_ref_{tag}.npz"), seed=seed, num_classes=nc, size=size, num_queries=nq,
                            x_seed=seed, logits=ref["pred_logits"].numpy(), boxes=ref["pred_boxes"].numpy())


def _random_layout(rng, n, page=(1200, 1600)):
    """Boxes that look like page elements: columns of stacked blocks plus a few strays/overlaps."""
    boxes = []
    ncol = int(rng.integers(1, 4))
    col_w = page[0] // ncol
    for c in range(ncol):
        y = int(rng.integers(20, 120))
        while y < page[1] - 80 and len(boxes) < n:
            h = int(rng.integers(20, 260))
            x1 = c * col_w + int(rng.integers(5, 60))
            x2 = (c + 1) * col_w - int(rng.integers(5, 60))
            if rng.random() < 0.25:  # short block
                x2 = x1 + int(rng.integers(40, max(41, (x2 - x1) // 2)))
            boxes.append([x1, y, max(x1 + 10, x2), y + h])
            y += h + int(rng.integers(-10, 60))
    while len(boxes) < n:
        x1, y1 = int(rng.integers(0, page[0] - 50)), int(rng.integers(0, page[1] - 50))
        boxes.append([x1, y1, x1 + int(rng.integers(10, 400)), y1 + int(rng.integers(10, 200))])
    rng.shuffle(boxes)
    return [list(map(int, b)) for b in boxes[:n]]


def pin_host_logic():
    """Golden answers of the reference's pure-Python stages (reading order, containment filters,
    table cell grid) on seeded random layouts -> tests/golden/host_logic.json."""
    import json
    from types import SimpleNamespace

    ro = ref_import("yomitoku.reading_order")
    misc = ref_import("yomitoku.utils.misc")

    class El(SimpleNamespace):
        def dict(self):
            return {"box": self.box, "order": self.order}

    rng = np.random.default_rng(2024)
    cases = []
    for k in range(240):
        n = int(rng.integers(2, 40))
        boxes = _random_layout(rng, n)
        direction = ["top2bottom", "right2left", "left2right"][k % 3]
        els = [El(box=b, order=0) for b in boxes]
        ro.prediction_reading_order(els, direction)
        cases.append({"direction": direction, "boxes": boxes, "order": [int(e.order) for e in els]})
    pairs = []
    for _ in range(300):
        a = _random_layout(rng, 1)[0]
        b = [a[0] + int(rng.integers(-30, 30)), a[1] + int(rng.integers(-30, 30)), a[2] + int(rng.integers(-30, 30)),
             a[3] + int(rng.integers(-30, 30))]
        if b[2] <= b[0] or b[3] <= b[1]:
            continue
        ratio, inter = misc.calc_overlap_ratio(a, b)
        pairs.append({"a": a, "b": b, "ratio": float(ratio), "inter": inter, "contained": bool(misc.is_contained(a, b)),
                      "ih": bool(misc.is_intersected_horizontal(a, b)), "iv": bool(misc.is_intersected_vertical(a, b))})
    with open(os.path.join(GOLDEN, "host_logic.json"), "w") as f:
        json.dump({"reading_order": cases, "pairs": pairs}, f)
    print(f"[host] wrote {len(cases)} reading-order cases, {len(pairs)} box pairs")


def _ref_document_analyzer():
    """Import the reference's document_analyzer.py with its heavy siblings (module classes that pull
    cv2 / onnx / torchvision) replaced by empty stand-ins: only the pure aggregation code is used."""
    import sys
    import types

    from ._refstubs import install_stubs

    install_stubs()
    for name, attrs in (
        ("yomitoku.text_detector", {"TextDetector": object}),
        ("yomitoku.text_recognizer", {"TextRecognizer": object}),
        ("yomitoku.layout_analyzer", {"LayoutAnalyzer": object}),
        ("yomitoku.utils.visualizer", {"det_visualizer": None, "reading_order_visualizer": None}),
        ("yomitoku.export", {"export_csv": None, "export_html": None, "export_markdown": None, "export_json": None}),
    ):
        if name not in sys.modules:
            m = types.ModuleType(name)
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    if not hasattr(sys.modules["omegaconf"], "OmegaConf"):
        sys.modules["omegaconf"].OmegaConf = object
    for pkg in ("yomitoku.schemas",):
        if pkg in sys.modules and not hasattr(sys.modules[pkg], "__file__"):
            del sys.modules[pkg]
    return ref_import("yomitoku.document_analyzer")


def _random_page_results(rng, da, page=(1200, 1600)):
    """Synthetic OCR + layout results shaped like a real page: text lines inside paragraph / cell /
    figure boxes plus strays, a ruled table with a span, vertical lines, page header / footer."""
    import types

    sch = __import__("yomitoku.schemas", fromlist=["x"])
    words = []

    def add_line(x1, y1, x2, y2, vertical=False):
        pts = [[x1, y1], [x2, y1], [x2, y2], [x1, y2]]
        text = "".join(rng.choice(list("あいうカキク漢字abc 12"), size=int(rng.integers(1, 9))))
        if rng.random() < 0.15:
            text = "".join(rng.choice(list("ふりがなカタ"), size=int(rng.integers(1, 5))))
        words.append(sch.WordPrediction(points=pts, content=text, direction="vertical" if vertical else "horizontal",
                                        rec_score=float(rng.random()), det_score=float(rng.random())))

    paragraphs, figures, tables = [], [], []
    y = 40
    for k in range(int(rng.integers(2, 7))):
        x1 = int(rng.integers(30, 200))
        x2 = int(rng.integers(600, 1150))
        nl = int(rng.integers(1, 6))
        lh = int(rng.integers(14, 40))
        role = [None, None, "section_headings", "page_header", "page_footer"][int(rng.integers(0, 5))]
        paragraphs.append(sch.Element(id=None, box=[x1 - 5, y - 4, x2 + 5, y + nl * (lh + 6)], score=0.9, role=role,
                                      contents=None))
        for i in range(nl):
            add_line(x1, y + i * (lh + 6), int(rng.integers(x1 + 60, x2)), y + i * (lh + 6) + lh)
            if rng.random() < 0.3:  # furigana-sized line above
                add_line(x1 + 10, y + i * (lh + 6) - 7, x1 + 60, y + i * (lh + 6) - 1)
        y += nl * (lh + 6) + int(rng.integers(20, 80))
    if rng.random() < 0.8:  # a table
        tx, ty = int(rng.integers(40, 200)), y
        nr, ncol = int(rng.integers(2, 5)), int(rng.integers(2, 5))
        cw, ch = int(rng.integers(90, 220)), int(rng.integers(36, 70))
        rows = [sch.TableLineSchema(box=[tx, ty + r * ch, tx + ncol * cw, ty + (r + 1) * ch], score=0.9) for r in range(nr)]
        cols = [sch.TableLineSchema(box=[tx + c * cw, ty, tx + (c + 1) * cw, ty + nr * ch], score=0.9) for c in range(ncol)]
        cells = [sch.TableCellSchema(col=c + 1, row=r + 1, col_span=1, row_span=1,
                                     box=[tx + c * cw, ty + r * ch, tx + (c + 1) * cw, ty + (r + 1) * ch], contents=None)
                 for r in range(nr) for c in range(ncol)]
        tables.append(sch.TableStructureRecognizerSchema(box=[tx, ty, tx + ncol * cw, ty + nr * ch], n_row=nr, n_col=ncol,
                                                         rows=rows, cols=cols, spans=[], cells=cells, order=0))
        for r in range(nr):
            for c in range(ncol):
                if rng.random() < 0.8:
                    add_line(tx + c * cw + 6, ty + r * ch + 8, tx + (c + 1) * cw - int(rng.integers(6, 40)), ty + r * ch + 30)
        if rng.random() < 0.5:  # a line spanning two cells
            add_line(tx + 10, ty + 10, tx + 2 * cw - 10, ty + 32)
        y += nr * ch + 40
    if rng.random() < 0.6:  # figure with a caption inside
        fx, fy = int(rng.integers(50, 500)), y
        figures.append(sch.Element(id=None, box=[fx, fy, fx + 400, fy + 250], score=0.8, role=None, contents=None))
        add_line(fx + 20, fy + 200, fx + 300, fy + 225)
        y += 290
    for _ in range(int(rng.integers(0, 4))):  # vertical text column + strays
        vx = int(rng.integers(900, 1150))
        add_line(vx, 100, vx + 30, int(rng.integers(300, 900)), vertical=True)
    for _ in range(int(rng.integers(0, 4))):
        sx, sy = int(rng.integers(0, 1000)), int(rng.integers(0, 1500))
        add_line(sx, sy, sx + int(rng.integers(20, 180)), sy + int(rng.integers(12, 40)))
    order = rng.permutation(len(words)).tolist()
    words = [words[i] for i in order]
    ocr = sch.OCRSchema(words=words)
    layout = sch.LayoutAnalyzerSchema(paragraphs=paragraphs, tables=tables, figures=figures)
    return ocr, layout


def pin_aggregate():
    """Golden answers of the reference's DocumentAnalyzer.aggregate / _split_text_across_cells on seeded
    synthetic page results -> tests/golden/aggregate.json (inputs and outputs as plain JSON)."""
    import json
    from types import SimpleNamespace

    da = _ref_document_analyzer()
    sch = __import__("yomitoku.schemas", fromlist=["x"])
    rng = np.random.default_rng(77)
    cases = []
    for k in range(60):
        ocr, layout = _random_page_results(rng, da)
        opts = dict(ignore_ruby=bool(k % 3 == 0), ruby_threshold=2.0, ignore_meta=bool(k % 4 == 1),
                    reading_order=["auto", "auto", "left2right", "top2bottom", "right2left"][k % 5])
        inp = {"ocr": ocr.model_dump(), "layout": layout.model_dump(), "opts": opts}
        det = sch.TextDetectorSchema(points=[w.points for w in ocr.words], scores=[w.det_score for w in ocr.words])
        split = da._split_text_across_cells(det.model_copy(deep=True), layout.model_copy(deep=True))
        self = SimpleNamespace(img=None, **opts)
        out = da.DocumentAnalyzer.aggregate(self, ocr.model_copy(deep=True), layout.model_copy(deep=True))
        res = sch.DocumentAnalyzerSchema(**out)
        cases.append({"input": inp, "output": res.model_dump(), "split": split.model_dump()})
    with open(os.path.join(GOLDEN, "aggregate.json"), "w") as f:
        json.dump(cases, f, ensure_ascii=False)
    print(f"[aggregate] wrote {len(cases)} cases; paragraphs in case 0: {len(cases[0]['output']['paragraphs'])}")


def _ref_functions(relpath, names, env):
    """Compile the named top-level functions of a reference file - its own code, lifted by ast so the module's
    heavy imports (cv2, onnx, configs ...) never run - into `env` (the reference's utils.misc helpers)."""
    import ast

    from ._refstubs import REF_SRC

    path = os.path.join(REF_SRC, "yomitoku", relpath)
    tree = ast.parse(open(path, encoding="utf-8").read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), (relpath, names)
    ns = dict(env)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def pin_filters():
    """Layout / table box filters and the RT-DETR post-processor: the reference's own functions on seeded
    random detections -> tests/golden/filters.json."""
    import copy
    import json
    import sys
    import types

    import torch

    misc = ref_import("yomitoku.utils.misc")
    env = {k: getattr(misc, k) for k in ("is_contained", "calc_intersection", "filter_by_flag")}
    within, across = _ref_functions("layout_parser.py", ["filter_contained_rectangles_within_category",
                                                         "filter_contained_rectangles_across_categories"], env)
    cells_fn, span_fn = _ref_functions("table_structure_recognizer.py", ["extract_cells", "filter_contained_cells_within_spancell"], env)
    rng = np.random.default_rng(4242)

    def elements(n, nest=0.4):
        out = []
        for b in _random_layout(rng, n):
            out.append({"box": b, "score": float(rng.random()), "role": None})
            if rng.random() < nest:  # a box nested in / nearly equal to the previous one
                d = rng.integers(0, 12, size=4)
                out.append({"box": [int(b[0] + d[0]), int(b[1] + d[1]), int(b[2] - d[2]), int(b[3] - d[3])], "score": float(rng.random()),
                            "role": None})
        return [e for e in out if e["box"][2] > e["box"][0] and e["box"][3] > e["box"][1]]

    layout_cases = []
    for _ in range(60):
        cat = {"paragraphs": elements(int(rng.integers(0, 14))), "tables": elements(int(rng.integers(0, 4)), 0.2),
               "figures": elements(int(rng.integers(0, 4)), 0.2)}
        inp = copy.deepcopy(cat)
        a = within(copy.deepcopy(cat))
        b = across(copy.deepcopy(a), "tables", "paragraphs")
        c = across(copy.deepcopy(b), "figures", "paragraphs")
        layout_cases.append({"input": inp, "within": a, "after_tables": b, "after_figures": c})
    table_cases = []
    for _ in range(60):
        x1, y1 = int(rng.integers(0, 300)), int(rng.integers(0, 300))
        w, h = int(rng.integers(200, 900)), int(rng.integers(120, 600))
        nr, nc = int(rng.integers(1, 8)), int(rng.integers(1, 7))
        ys = np.sort(rng.integers(y1, y1 + h, size=nr + 1))
        xs = np.sort(rng.integers(x1, x1 + w, size=nc + 1))
        rows = [[x1, int(ys[i]) - int(rng.integers(0, 4)), x1 + w, int(ys[i + 1]) + int(rng.integers(0, 4))] for i in range(nr)]
        cols = [[int(xs[j]) - int(rng.integers(0, 4)), y1, int(xs[j + 1]) + int(rng.integers(0, 4)), y1 + h] for j in range(nc)]
        spans = []
        for _s in range(int(rng.integers(0, 3))):
            i0, j0 = int(rng.integers(0, nr)), int(rng.integers(0, nc))
            i1, j1 = int(rng.integers(i0, nr)), int(rng.integers(j0, nc))
            spans.append([int(xs[j0]) - 2, int(ys[i0]) - 2, int(xs[j1 + 1]) + 2, int(ys[i1 + 1]) + 2])
        cells = cells_fn(rows, cols)
        merged = span_fn(copy.deepcopy(cells), spans)
        table_cases.append({"rows": rows, "cols": cols, "spans": spans, "cells": cells, "merged": merged})
    # RT-DETR post-processor (torchvision.ops.box_convert is the only torchvision call: cxcywh -> xyxy)
    tv = sys.modules["torchvision"]
    if not hasattr(tv, "ops"):
        def box_convert(boxes, in_fmt, out_fmt):
            assert (in_fmt, out_fmt) == ("cxcywh", "xyxy")
            cx, cy, w, h = boxes.unbind(-1)
            return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)

        tv.ops = types.SimpleNamespace(box_convert=box_convert)
    pp = ref_import("yomitoku.postprocessor.rtdetr_postprocessor")
    post_cases = []
    g = torch.Generator().manual_seed(99)
    for nc, thr in ((6, 0.5), (3, 0.4), (6, 0.3)):
        post = pp.RTDETRPostProcessor(num_classes=nc, num_top_queries=300)
        logits = torch.randn(1, 300, nc, generator=g) * 2.0 - 4.0
        boxes = torch.rand(1, 300, 4, generator=g) * torch.tensor([1.0, 1.0, 0.6, 0.6]) + torch.tensor([0.0, 0.0, 0.02, 0.02])
        size = (int(torch.randint(300, 1700, (1,), generator=g)), int(torch.randint(300, 1700, (1,), generator=g)))  # (w, h)
        res = post({"pred_logits": logits.clone(), "pred_boxes": boxes.clone()}, torch.tensor([size]), thr)[0]
        post_cases.append({"num_classes": nc, "threshold": thr, "size_wh": list(size), "logits": logits[0].tolist(),
                           "boxes": boxes[0].tolist(), "labels": res["labels"].tolist(), "out_boxes": res["boxes"].tolist(),
                           "scores": res["scores"].tolist()})
    with open(os.path.join(GOLDEN, "filters.json"), "w") as f:
        json.dump({"layout": layout_cases, "table": table_cases, "post": post_cases}, f)
    print(f"[filters] wrote {len(layout_cases)} layout, {len(table_cases)} table, {len(post_cases)} post-processor cases; "
          f"kept detections: {[len(c['labels']) for c in post_cases]}")


def pin_cells():
    """Table cell detector post-processing (table_cell_detector.py:41-192,339-489): the reference's own functions and
    CellDetector methods, lifted by ast, on seeded random detections -> tests/golden/cells.json.  The one piece that
    cannot run here is find_holes_as_rects (OpenCV): the product's C++ restatement stands in for it on BOTH sides, so
    this golden pins everything around it (class filters, whole-crop rejection, hole adoption by adjacency, roles, ids,
    noise-cell removal, kv / grid regions); the hole finder itself has its own second-source test."""
    import json
    import math
    import types

    import torch

    from yomitoku_amd import schemas as my_schemas
    from yomitoku_amd.table_cell_detector import find_holes_as_rects

    misc_names = ["filter_by_flag", "calc_intersection", "calc_overlap_ratio", "is_contained", "calc_iou", "clamp",
                  "point_to_segment_distance", "right_edge_to_left_edge_dist", "top_edge_to_bottom_edge_dist", "overlap_interval",
                  "point_distance", "is_right_adjacent", "is_bottom_adjacent"]
    env = {"math": math}
    env.update(zip(misc_names, _ref_functions("utils/misc.py", misc_names, env)))
    env = dict(env)
    # re-lift so that the helpers see each other (they are looked up in the namespace they were compiled in)
    env.update(zip(misc_names, _ref_functions("utils/misc.py", misc_names, env)))
    fn_names = ["filter_contained_rectangles_with_category", "filter_contained_rectangles_across_categories", "choose_role",
                "calc_adjacent_holes_to_cells"]
    env.update(zip(fn_names, _ref_functions("table_cell_detector.py", fn_names, env)))
    env.update(find_holes_as_rects=find_holes_as_rects, torch=torch, CellSchema=my_schemas.CellSchema,
               RegionSchema=my_schemas.RegionSchema)
    tv = sys.modules.get("torchvision") or types.ModuleType("torchvision")
    if not hasattr(tv, "ops"):
        def box_convert(boxes, in_fmt, out_fmt):
            assert (in_fmt, out_fmt) == ("cxcywh", "xyxy")
            cx, cy, w, h = boxes.unbind(-1)
            return torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], dim=-1)

        tv.ops = types.SimpleNamespace(box_convert=box_convert)
        sys.modules.setdefault("torchvision", tv)
    pp = ref_import("yomitoku.postprocessor.rtdetr_postprocessor")
    m_names = ["is_fully_contained", "postprocess", "remove_noise_cells", "extract_cell_elements"]
    methods = dict(zip(m_names, _ref_methods("table_cell_detector.py", "CellDetector", m_names, env)))
    categories = ["table", "cell", "header", "empty", "kv_item", "grid"]
    det = types.SimpleNamespace(device="cpu", thresh_score=0.5, label_mapper=dict(enumerate(categories)),
                                postprocessor=pp.RTDETRPostProcessor(num_classes=6, num_top_queries=1500))
    for name, fn in methods.items():
        setattr(det, name, types.MethodType(fn, det))
    rng = np.random.default_rng(515)
    cases = []
    for k in range(40):
        w, h = int(rng.integers(200, 900)), int(rng.integers(120, 600))
        ox, oy = int(rng.integers(0, 300)), int(rng.integers(0, 300))
        rows, cols = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        boxes, logits = [], []
        for r in range(rows):
            for c in range(cols):
                if rng.random() < 0.18:
                    continue  # a missing cell: a hole candidate
                cx, cy = (c + 0.5) / cols + rng.normal(0, 0.004), (r + 0.5) / rows + rng.normal(0, 0.004)
                bw, bh = 1.0 / cols * rng.uniform(0.9, 1.0), 1.0 / rows * rng.uniform(0.9, 1.0)
                boxes.append([cx, cy, bw, bh])
                lg = np.full(6, -6.0)
                lg[int(rng.choice([1, 1, 1, 2, 3]))] = rng.uniform(0.5, 4.0)
                logits.append(lg)
                if rng.random() < 0.15:  # a duplicate, slightly smaller or larger, maybe of another class
                    boxes.append([cx, cy, bw * rng.uniform(0.8, 1.05), bh * rng.uniform(0.8, 1.05)])
                    lg2 = np.full(6, -6.0)
                    lg2[int(rng.choice([1, 2, 3]))] = rng.uniform(0.2, 3.0)
                    logits.append(lg2)
        for cls in (0, 4, 5):  # whole-crop table box, kv_item / grid regions
            if rng.random() < 0.7:
                boxes.append([0.5, 0.5, rng.uniform(0.93, 1.0), rng.uniform(0.93, 1.0)])
                lg = np.full(6, -6.0)
                lg[cls] = rng.uniform(0.5, 3.0)
                logits.append(lg)
        n = len(boxes)
        pad = 1500 - n
        bx = np.concatenate([np.asarray(boxes, dtype=np.float32).reshape(n, 4), rng.random((pad, 4)).astype(np.float32) * 0.3 + 0.1])
        lg = np.concatenate([np.asarray(logits, dtype=np.float32).reshape(n, 6), np.full((pad, 6), -8.0, dtype=np.float32)])
        preds = {"pred_logits": torch.from_numpy(lg)[None], "pred_boxes": torch.from_numpy(bx)[None]}
        table_box = [ox, oy, ox + w, oy + h]
        cells, kv, grid = det.postprocess(preds, {"size": (h, w), "offset": (ox, oy)}, table_box)
        cases.append({"size": [h, w], "offset": [ox, oy], "table_box": table_box, "n_real": n, "logits": lg[:n].tolist(),
                      "boxes": bx[:n].tolist(), "cells": [c.model_dump() for c in cells], "kv": [r.model_dump() for r in kv],
                      "grid": [r.model_dump() for r in grid]})
    with open(os.path.join(GOLDEN, "cells.json"), "w") as f:
        json.dump({"cases": cases}, f)
    roles = {}
    for c in cases:
        for cell in c["cells"]:
            roles[cell["role"]] = roles.get(cell["role"], 0) + 1
    print(f"[cells] wrote {len(cases)} cases; cells by role: {roles}; kv {sum(len(c['kv']) for c in cases)}, grid {sum(len(c['grid']) for c in cases)}")


def _export_documents(n=12, seed=808):
    """Seeded DocumentAnalyzerSchema objects with awkward text (markdown / HTML specials, URLs, line breaks, spans)."""
    from yomitoku_amd.schemas import (DocumentAnalyzerSchema, FigureSchema, ParagraphSchema, TableCellSchema, TableLineSchema,
                                      TableStructureRecognizerSchema)

    rng = np.random.default_rng(seed)
    words = ["請求書", "total*", "a|b", "#1 [x](y)", "see https://example.com/a?b=1&c=<2>", "line1\nline2", "`code`", "5 < 7 & 9 > 8",
             "~del~", "+1-2", "{k}", "plain text", "表 3", "!bang", "multi\nline\ntext", ""]

    def text():
        return " ".join(words[int(i)] for i in rng.integers(0, len(words), int(rng.integers(1, 4))))

    docs = []
    for _ in range(n):
        order = 0
        paragraphs, tables, figures = [], [], []
        for _p in range(int(rng.integers(1, 6))):
            role = [None, None, "section_headings", "page_header"][int(rng.integers(0, 4))]
            paragraphs.append(ParagraphSchema(box=[0, order * 10, 100, order * 10 + 8], contents=text(), direction="horizontal", order=order, role=role))
            order += 1
        for _t in range(int(rng.integers(0, 3))):
            n_row, n_col = int(rng.integers(1, 5)), int(rng.integers(1, 5))
            taken, cells = set(), []
            for r in range(1, n_row + 1):
                for c in range(1, n_col + 1):
                    if (r, c) in taken:
                        continue
                    rs = int(rng.integers(1, min(2, n_row - r + 1) + 1))
                    cs = int(rng.integers(1, min(2, n_col - c + 1) + 1))
                    if any((r + a, c + b) in taken for a in range(rs) for b in range(cs)):
                        rs = cs = 1
                    taken.update((r + a, c + b) for a in range(rs) for b in range(cs))
                    cells.append(TableCellSchema(col=c, row=r, col_span=cs, row_span=rs, box=[c, r, c + cs, r + rs], contents=text()))
            line = TableLineSchema(box=[0, 0, 1, 1], score=0.9)
            tables.append(TableStructureRecognizerSchema(box=[0, order * 10, 100, order * 10 + 8], n_row=n_row, n_col=n_col, rows=[line] * n_row,
                                                         cols=[line] * n_col, spans=[], cells=cells, order=order))
            order += 1
        for _f in range(int(rng.integers(0, 2))):
            inner = [ParagraphSchema(box=[1, 1, 5, 5], contents=text(), direction="horizontal", order=k, role=None) for k in range(2)]
            figures.append(FigureSchema(box=[2, 2, 30, 30], order=order, paragraphs=inner, direction="horizontal"))
            order += 1
        docs.append(DocumentAnalyzerSchema(paragraphs=paragraphs, tables=tables, words=[], figures=figures))
    return docs


def pin_export():
    """Exporter text conversions (export/export_csv.py, export_markdown.py, export_html.py): the reference's own functions,
    lifted by ast, on seeded documents -> tests/golden/export.json.  Figures are not written (export_figure=False): that
    part needs cv2.imencode; convert_html's lxml pretty-printing is not reproduced either - its ELEMENTS are pinned."""
    import csv as csv_mod
    import json
    import re as re_mod
    from html import escape

    env_csv = {"csv": csv_mod, "os": os, "save_image": None}
    t2c, p2c, conv_csv = _ref_functions("export/export_csv.py", ["table_to_csv", "paragraph_to_csv", "convert_csv"], env_csv)
    env_csv.update(table_to_csv=t2c, paragraph_to_csv=p2c)
    t2c, p2c, conv_csv = _ref_functions("export/export_csv.py", ["table_to_csv", "paragraph_to_csv", "convert_csv"], env_csv)
    md_names = ["escape_markdown_special_chars", "paragraph_to_md", "table_to_md", "convert_markdown"]
    env_md = {"re": re_mod, "os": os}
    env_md.update(zip(md_names, _ref_functions("export/export_markdown.py", md_names, env_md)))
    md = dict(zip(md_names, _ref_functions("export/export_markdown.py", md_names, env_md)))
    html_names = ["convert_text_to_html", "add_td_tag", "add_table_tag", "add_tr_tag", "add_p_tag", "add_h1_tag", "table_to_html",
                  "paragraph_to_html"]
    env_html = {"re": re_mod, "os": os, "escape": escape}
    env_html.update(zip(html_names, _ref_functions("export/export_html.py", html_names, env_html)))
    hf = dict(zip(html_names, _ref_functions("export/export_html.py", html_names, env_html)))
    cases = []
    for doc in _export_documents():
        for ilb in (False, True):
            for letter in (False, True):
                csv_el = conv_csv(doc.model_copy(deep=True), "out.csv", ilb, None, False, letter, "figures")
                md_text, md_el = md["convert_markdown"](doc.model_copy(deep=True), "out.md", ilb, None, letter, False, 200, "figures")
                cases.append({"doc": doc.model_dump(), "ignore_line_break": ilb, "figure_letter": letter,
                              "csv": [{"type": e["type"], "element": e["element"], "order": e["order"]} for e in csv_el],
                              "markdown": md_text,
                              "html_tables": [hf["table_to_html"](t, ilb)["html"] for t in doc.tables],
                              "html_paragraphs": [hf["paragraph_to_html"](p_, ilb)["html"] for p_ in doc.paragraphs]})
    with open(os.path.join(GOLDEN, "export.json"), "w") as f:
        json.dump({"cases": cases}, f, ensure_ascii=False)
    print(f"[export] wrote {len(cases)} cases")


def _pdf_documents(n=10, seed=909, page=(1000, 1400)):
    """Seeded pages for the searchable-PDF text layer: horizontal and vertical paragraphs, a table, a figure, words inside /
    outside / across containers, contents with ASCII, digits, half-width katakana (with voiced marks), yen sign, spaces."""
    from yomitoku_amd.schemas import (DocumentAnalyzerSchema, FigureSchema, ParagraphSchema, TableCellSchema, TableLineSchema,
                                      TableStructureRecognizerSchema, WordPrediction)

    rng = np.random.default_rng(seed)
    W, H = page
    texts = ["請求書", "total 123", "ｶﾞｷﾞｸﾞ ﾊﾟﾋﾟ", "ｱｲｳｴｵ｡｢｣､･ｰ", "¥1,200", "a·b", "東京都千代田区1-2-3", "ABC abc 012", "縦書きのテキスト", "ﾞﾟ", "",
              "x", "〒100-0001", "Tel: 03(1234)5678", "表 3", "ｳﾞｧｲｵﾘﾝ", "１２３ＡＢＣ", "~!@#$%^&*()_+{}|:<>?"]

    def word(box, direction):
        x1, y1, x2, y2 = box
        jit = lambda: int(rng.integers(-2, 3))  # a quadrangle, not a rectangle: the hull of its corners is what counts
        pts = [[x1 + jit(), y1 + jit()], [x2 + jit(), y1 + jit()], [x2 + jit(), y2 + jit()], [x1 + jit(), y2 + jit()]]
        return WordPrediction(points=pts, content=texts[int(rng.integers(0, len(texts)))], direction=direction,
                              rec_score=0.9, det_score=0.9)

    docs = []
    for _ in range(n):
        order = 0
        paragraphs, tables, figures, words = [], [], [], []
        for _p in range(int(rng.integers(2, 6))):
            vertical = bool(rng.integers(0, 3) == 0)
            x, y = int(rng.integers(20, W - 320)), int(rng.integers(20, H - 320))
            w, h = (int(rng.integers(60, 160)), int(rng.integers(200, 300))) if vertical else (int(rng.integers(200, 300)), int(rng.integers(60, 160)))
            direction = "vertical" if vertical else "horizontal"
            paragraphs.append(ParagraphSchema(box=[x, y, x + w, y + h], contents="", direction=direction, order=order, role=None))
            order += 1
            for k in range(int(rng.integers(1, 4))):  # lines of the paragraph
                if vertical:
                    lx = x + w - (k + 1) * (w // 3)
                    words.append(word([lx, y + 2, lx + w // 3 - 4, y + h - int(rng.integers(2, 60))], direction))
                else:
                    ly = y + k * (h // 3)
                    words.append(word([x + 2, ly, x + w - int(rng.integers(2, 60)), ly + h // 3 - 4], direction))
        # words outside every container, straddling one, of zero height
        words.append(word([W - 200, H - 40, W - 20, H - 12], "horizontal"))
        p0 = paragraphs[0].box
        words.append(word([p0[0] - 30, p0[1] + 5, p0[0] + 60, p0[1] + 35], "horizontal"))
        words.append(WordPrediction(points=[[p0[0] + 5, p0[1] + 5], [p0[0] + 80, p0[1] + 5], [p0[0] + 80, p0[1] + 5], [p0[0] + 5, p0[1] + 5]],
                                    content="flat", direction="horizontal", rec_score=0.9, det_score=0.9))
        for _t in range(int(rng.integers(0, 2))):
            tx, ty = int(rng.integers(20, W - 420)), int(rng.integers(20, H - 220))
            n_row, n_col = int(rng.integers(1, 4)), int(rng.integers(1, 4))
            cw, ch = 400 // n_col, 200 // n_row
            cells = []
            for r in range(n_row, 0, -1):  # cells listed backwards: the sub-order has to sort them
                for c in range(n_col, 0, -1):
                    box = [tx + (c - 1) * cw, ty + (r - 1) * ch, tx + c * cw, ty + r * ch]
                    cells.append(TableCellSchema(col=c, row=r, col_span=1, row_span=1, box=box, contents=""))
                    words.append(word([box[0] + 3, box[1] + 3, box[2] - 3, box[3] - 3], "horizontal"))
            line = TableLineSchema(box=[0, 0, 1, 1], score=0.9)
            tables.append(TableStructureRecognizerSchema(box=[tx, ty, tx + 400, ty + 200], n_row=n_row, n_col=n_col, rows=[line] * n_row,
                                                         cols=[line] * n_col, spans=[], cells=cells, order=order))
            order += 1
        for _f in range(int(rng.integers(0, 2))):
            fx, fy = int(rng.integers(20, W - 320)), int(rng.integers(20, H - 220))
            inner = []
            for k in range(2):
                box = [fx + 10, fy + 10 + k * 80, fx + 250, fy + 70 + k * 80]
                inner.append(ParagraphSchema(box=box, contents="", direction="horizontal", order=k, role=None))
                words.append(word([box[0] + 2, box[1] + 2, box[2] - 2, box[3] - 2], "horizontal"))
            figures.append(FigureSchema(box=[fx, fy, fx + 300, fy + 200], order=order, paragraphs=inner, direction="horizontal"))
            order += 1
        perm = rng.permutation(len(words))
        docs.append(DocumentAnalyzerSchema(paragraphs=paragraphs, tables=tables, words=[words[int(k)] for k in perm], figures=figures))
    return docs


def pin_searchable_pdf():
    """utils/searchable_pdf.py: create_searchable_pdf and its helpers, the reference's own code lifted by ast, run against a
    canvas that RECORDS what reportlab would have been asked to draw -> tests/golden/searchable_pdf.json.  What this pins:
    which words are written, in which order, with which font size, at which text matrix.  What it cannot: reportlab's
    stringWidth of the MPLUS face and jaconv.h2z are not installed - both sides use yomitoku_amd's width model and h2z."""
    import json
    import tempfile
    from io import BytesIO
    from types import SimpleNamespace
    from typing import List, Optional

    from PIL import Image

    from yomitoku_amd.schemas import DocumentAnalyzerSchema
    from yomitoku_amd.utils.searchable_pdf import h2z, string_width

    calc_intersection, calc_overlap_ratio, is_contained = _ref_functions("utils/misc.py", ["calc_intersection", "calc_overlap_ratio", "is_contained"], {})
    env_misc = {"calc_intersection": calc_intersection}
    (calc_overlap_ratio,) = _ref_functions("utils/misc.py", ["calc_overlap_ratio"], env_misc)
    (is_contained,) = _ref_functions("utils/misc.py", ["is_contained"], {"calc_overlap_ratio": calc_overlap_ratio})

    pages = []

    class RecordingCanvas:
        def __init__(self, packet):
            self.ctm, self.stack, self.ops = np.eye(3), [], []

        def _cm(self, m):
            self.ctm = np.array(m, dtype=np.float64) @ self.ctm

        def setPageSize(self, size):
            self.size = [int(size[0]), int(size[1])]

        def drawImage(self, path, x, y, width, height):
            with Image.open(path) as im:
                assert im.format == "JPEG" and im.size == (width, height) and (x, y) == (0, 0)

        def setFillColor(self, color):
            assert color == ("color", 1, 1, 1, 0)

        def setFont(self, name, size):
            self.ops.append(["font", float(size)])

        def saveState(self):
            self.stack.append(self.ctm.copy())

        def restoreState(self):
            self.ctm = self.stack.pop()

        def translate(self, dx, dy):
            self._cm([[1, 0, 0], [0, 1, 0], [float(dx), float(dy), 1]])

        def rotate(self, theta):
            c, s = np.cos(np.radians(theta)), np.sin(np.radians(theta))
            self._cm([[c, s, 0], [-s, c, 0], [0, 0, 1]])

        def drawString(self, x, y, text):
            m = np.array([[1, 0, 0], [0, 1, 0], [float(x), float(y), 1]]) @ self.ctm
            self.ops.append(["text"] + [round(float(v), 9) + 0.0 for v in (m[0, 0], m[0, 1], m[1, 0], m[1, 1], m[2, 0], m[2, 1])] + [text])

        def showPage(self):
            assert not self.stack
            pages.append({"size": self.size, "ops": self.ops})
            self.ctm, self.ops = np.eye(3), []

        def save(self):
            pass

    env = {"np": np, "os": os, "BytesIO": BytesIO, "Image": Image, "List": List, "Optional": Optional,
           "DocumentAnalyzerSchema": DocumentAnalyzerSchema, "is_contained": is_contained,
           "stringWidth": lambda text, font, size: string_width(text, size),
           "jaconv": SimpleNamespace(h2z=lambda text, kana, ascii, digit: h2z(text)),
           "pdfmetrics": SimpleNamespace(registerFont=lambda font: None), "TTFont": lambda name, path: (name, path),
           "canvas": SimpleNamespace(Canvas=RecordingCanvas), "Color": lambda r, g, b, alpha: ("color", r, g, b, alpha),
           "FONT_PATH": "unused.ttf",
           "IMAGE_QUALITY_PRESETS": {"high": {"max_long_side": None, "jpeg_quality": 85}, "middle": {"max_long_side": 2000, "jpeg_quality": 80},
                                     "low": {"max_long_side": 1500, "jpeg_quality": 60}}}
    names = ["_poly2rect", "_calc_font_size", "to_full_width", "create_searchable_pdf"]
    env.update(zip(names, _ref_functions("utils/searchable_pdf.py", names, env)))
    create = _ref_functions("utils/searchable_pdf.py", names, env)[3]
    docs = _pdf_documents()
    rng = np.random.default_rng(5)
    images = [Image.fromarray(rng.integers(0, 255, (1400, 1000, 3), dtype=np.uint8)) for _ in docs]
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)  # the reference writes tmp_<i>.png into the working directory
        try:
            create(images, docs, os.path.join(tmp, "out.pdf"))
        finally:
            os.chdir(cwd)
    assert len(pages) == len(docs)
    to_fw = _ref_functions("utils/searchable_pdf.py", names, env)[2]
    samples = ["ｶﾞｷﾞ ABC 012", "¥100·200", "ｱﾞ ﾞ ﾟ", "mixed 全角 ﾊﾝｶｸ", ""]
    out = {"pages": [{"doc": d.model_dump(), "size": p["size"], "ops": p["ops"]} for d, p in zip(docs, pages)],
           "to_full_width": [[s_, to_fw(s_)] for s_ in samples]}
    with open(os.path.join(GOLDEN, "searchable_pdf.json"), "w") as f:
        json.dump(out, f, ensure_ascii=False)
    n_text = sum(1 for p in pages for op in p["ops"] if op[0] == "text")
    n_rot = sum(1 for p in pages for op in p["ops"] if op[0] == "text" and op[2] == -1.0)
    print(f"[searchable_pdf] wrote {len(pages)} pages, {n_text} strings ({n_rot} turned characters of vertical words), "
          f"{sum(len(d.words) for d in docs)} words in")


REFERENCE_UNIT_TESTS = ["test_extract_paragraph_within_figure", "test_combile_flags", "test_judge_page_direction",
                        "test_extract_words_within_element", "test_is_vertical", "test_is_noise", "test_recursive_update",
                        "test_extract_words_within_table", "test_calc_overlap_words_on_lines", "test_correct_vertical_word_boxes",
                        "test_correct_horizontal_word_boxes", "test_split_text_across_cells"]
REFERENCE_UNIT_FUNCTIONS = ["extract_paragraph_within_figure", "combine_flags", "judge_page_direction", "extract_words_within_element",
                            "is_vertical", "is_noise", "recursive_update", "_extract_words_within_table", "_calc_overlap_words_on_lines",
                            "_correct_vertical_word_boxes", "_correct_horizontal_word_boxes", "_split_text_across_cells"]


def _plain(v):
    """Arguments and results of the recorded calls as JSON: pydantic models tagged with their class name, tuples and arrays
    tagged, everything else as it is."""
    if hasattr(v, "model_dump"):
        return {"__model__": type(v).__name__, "fields": v.model_dump()}
    if isinstance(v, np.ndarray):
        return {"__ndarray__": v.tolist()}
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, tuple):
        return {"__tuple__": [_plain(x) for x in v]}
    if isinstance(v, list):
        return [_plain(x) for x in v]
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    return v


def pin_reference_unit_tests():
    """The reference's OWN unit tests of the aggregation helpers (tests/test_document_analyzer.py:116-600), run here against
    the reference's own functions with every call recorded: (function, arguments before the call, result) ->
    tests/golden/reference_unit_cases.json.  tests/test_reference_unit_cases.py replays the calls on yomitoku_amd's
    functions.  The test bodies are lifted by ast (the file's imports - cv2, omegaconf, the model classes - never run) and
    their own asserts hold while recording, so the recorded results are the known answers the reference's authors wrote."""
    import ast
    import copy
    import json

    import pytest

    from ._refstubs import REF_SRC

    da = _ref_document_analyzer()
    sch = __import__("yomitoku.schemas", fromlist=["x"])
    records = []

    def recorder(name, fn, module="document_analyzer"):
        def wrapped(*args, **kwargs):
            before = (_plain(copy.deepcopy(args)), _plain(copy.deepcopy(kwargs)))
            out = fn(*args, **kwargs)
            records.append({"module": module, "fn": name, "args": before[0]["__tuple__"], "kwargs": before[1], "result": _plain(copy.deepcopy(out)),
                            "args_after": _plain(copy.deepcopy(args))["__tuple__"]})  # some helpers work in place
            return out
        return wrapped

    env = {name: recorder(name, getattr(da, name)) for name in REFERENCE_UNIT_FUNCTIONS}
    for cls in ("DocumentAnalyzerSchema", "ParagraphSchema", "FigureSchema", "TextDetectorSchema", "TableStructureRecognizerSchema",
                "TableLineSchema", "TableCellSchema", "WordPrediction"):
        env[cls] = getattr(sch, cls)
    env.update(pytest=pytest, np=np)
    path = os.path.join(os.path.dirname(REF_SRC), "tests", "test_document_analyzer.py")
    tree = ast.parse(open(path, encoding="utf-8").read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in REFERENCE_UNIT_TESTS]
    assert {n.name for n in body} == set(REFERENCE_UNIT_TESTS), sorted(set(REFERENCE_UNIT_TESTS) - {n.name for n in body})
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), env)
    per_test = {}
    for name in REFERENCE_UNIT_TESTS:
        n0 = len(records)
        env[name]()  # the reference's own asserts run against the reference's own functions
        per_test[name] = len(records) - n0

    # ---- tests/test_export.py:37-455: the converters of the four text exporters (their files import cv2 / lxml: lifted by ast)
    import csv as csv_mod
    import re as re_mod
    from html import escape

    env_csv = {"csv": csv_mod, "os": os, "save_image": None}
    env_csv.update(zip(["table_to_csv", "paragraph_to_csv"], _ref_functions("export/export_csv.py", ["table_to_csv", "paragraph_to_csv"], env_csv)))
    md_names = ["escape_markdown_special_chars", "paragraph_to_md", "table_to_md"]
    env_md = {"re": re_mod, "os": os}
    env_md.update(zip(md_names, _ref_functions("export/export_markdown.py", md_names, env_md)))
    env_md.update(zip(md_names, _ref_functions("export/export_markdown.py", md_names, env_md)))
    html_names = ["convert_text_to_html", "add_td_tag", "add_table_tag", "add_tr_tag", "add_p_tag", "add_h1_tag", "table_to_html", "paragraph_to_html"]
    env_html = {"re": re_mod, "os": os, "escape": escape}
    env_html.update(zip(html_names, _ref_functions("export/export_html.py", html_names, env_html)))
    env_html.update(zip(html_names, _ref_functions("export/export_html.py", html_names, env_html)))
    json_fns = dict(zip(["paragraph_to_json", "table_to_json"], _ref_functions("export/export_json.py", ["paragraph_to_json", "table_to_json"], {})))
    export_fns = {"table_to_csv": env_csv["table_to_csv"], "paragraph_to_csv": env_csv["paragraph_to_csv"],
                  **{k: env_md[k] for k in md_names}, "convert_text_to_html": env_html["convert_text_to_html"],
                  "table_to_html": env_html["table_to_html"], "paragraph_to_html": env_html["paragraph_to_html"], **json_fns}
    export_tests = ["test_convert_text_to_html", "test_table_to_html", "test_paragraph_to_html", "test_escape_markdown_special_chars",
                    "test_paragraph_to_md", "test_table_to_md", "test_table_to_csv", "test_paragraph_to_csv", "test_paragraph_to_json",
                    "test_table_to_json"]
    env2 = {name: recorder(name, fn, "export") for name, fn in export_fns.items()}
    for cls in ("DocumentAnalyzerSchema", "LayoutAnalyzerSchema", "LayoutParserSchema", "OCRSchema", "ParagraphSchema", "FigureSchema",
                "TableCellSchema", "TableLineSchema", "TableStructureRecognizerSchema", "TextDetectorSchema", "TextRecognizerSchema",
                "WordPrediction", "Element"):
        env2[cls] = getattr(sch, cls)
    env2.update(np=np, os=os, json=json)
    path2 = os.path.join(os.path.dirname(REF_SRC), "tests", "test_export.py")
    tree2 = ast.parse(open(path2, encoding="utf-8").read())
    body2 = [n for n in tree2.body if isinstance(n, ast.FunctionDef) and n.name in export_tests]
    assert {n.name for n in body2} == set(export_tests)
    exec(compile(ast.Module(body=body2, type_ignores=[]), path2, "exec"), env2)
    for name in export_tests:
        n0 = len(records)
        env2[name]()
        per_test[name] = len(records) - n0
    with open(os.path.join(GOLDEN, "reference_unit_cases.json"), "w") as f:
        json.dump({"source": "tests/test_document_analyzer.py and tests/test_export.py of the reference, run against the reference's own functions",
                   "calls_per_test": per_test, "calls": records}, f, ensure_ascii=False)
    print(f"[reference unit tests] {len(per_test)} tests, {len(records)} recorded calls: {per_test}")


def _ref_methods(relpath, cls, names, env):
    """Like _ref_functions for methods of a reference class: returned as plain functions taking `self` first."""
    import ast

    from ._refstubs import REF_SRC

    path = os.path.join(REF_SRC, "yomitoku", relpath)
    tree = ast.parse(open(path, encoding="utf-8").read())
    klass = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == cls)
    body = [n for n in klass.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), (relpath, cls, names)
    ns = dict(env)
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def pin_geometry():
    """Integer geometry and batching rules of the recogniser / detector front ends, answered by the reference's own
    functions (cv2.resize replaced by a stand-in that only reports the size it was asked for):
    resize_shortest_edge sizes, calc_resize_without_padding, validate_quads, _calc_source_levels,
    TextRecognizer._make_mini_batch compositions, ParseqTokenizer.decode -> tests/golden/geometry.json."""
    import json
    import types

    import torch
    import torch.nn.functional as F

    cv2 = types.SimpleNamespace(INTER_AREA=3, resize=lambda img, dsize, interpolation=None: np.zeros((dsize[1], dsize[0], 3), np.uint8))
    fenv = {"np": np, "cv2": cv2}
    rse, calc, validate = _ref_functions("data/functions.py", ["resize_shortest_edge", "calc_resize_without_padding", "validate_quads"], fenv)
    short_sides, levels_fn = _ref_functions("data/dataset.py", ["_quad_short_sides", "_calc_source_levels"], {"np": np})
    rng = np.random.default_rng(77)
    sizes = [(1600, 1200), (1200, 1600), (480, 640), (2100, 1500), (91, 38), (700, 2400), (32, 32), (33, 4000)]
    sizes += [(int(rng.integers(20, 5000)), int(rng.integers(20, 5000))) for _ in range(300)]
    resize_cases = []
    for h, w in sizes:
        out = rse(np.zeros((h, w, 3), np.uint8), 1280, 1600)
        resize_cases.append({"h": h, "w": w, "out": [int(out.shape[0]), int(out.shape[1])]})
    pad_cases = []
    for _ in range(400):
        h, w = int(rng.integers(1, 700)), int(rng.integers(1, 3000))
        nh, nw = calc(np.zeros((h, w, 3), np.uint8), (32, 800))
        pad_cases.append({"h": h, "w": w, "out": [int(nh), int(nw)]})
    quad_cases = []
    img = np.zeros((300, 400, 3), np.uint8)
    for _ in range(300):
        q = (rng.integers(-20, 430, size=(4, 2)) + rng.random((4, 2)) * (rng.random() < 0.3)).tolist()
        if rng.random() < 0.05:
            q = q[:3]
        quad_cases.append({"quad": q, "valid": validate(img, q) is True})
    level_cases = []
    for _ in range(40):
        n = int(rng.integers(1, 12))
        quads = []
        for _q in range(n):
            x, y = float(rng.integers(0, 500)), float(rng.integers(0, 500))
            w, h = float(rng.integers(2, 1200)), float(rng.integers(2, 600))
            quads.append([[x, y], [x + w, y + rng.integers(-3, 4)], [x + w, y + h], [x, y + h]])
        quads = np.asarray(quads, dtype=np.float64).tolist()
        level_cases.append({"quads": quads, "levels": [int(v) for v in levels_fn(quads, 32)]})
    mk, collate = _ref_methods("text_recognizer.py", "TextRecognizer", ["_make_mini_batch", "_collate"], {"torch": torch, "F": F, "np": np})
    batch_cases = []
    for k in range(60):
        n = int(rng.integers(1, 150))
        widths = (np.clip(rng.lognormal(np.log(120), 0.8, size=n), 16, 800).astype(int) // 8 * 8 + 8).tolist()
        dynamic = bool(k % 3 != 2)
        budget = [8000, None, 3000][k % 3] if dynamic else None
        max_bs = [64, None, 16][(k // 3) % 3]
        data = types.SimpleNamespace(batch_size=[10, 128, 7][k % 3], width_budget=budget, max_batch_size=max_bs)
        fake = types.SimpleNamespace(_cfg=types.SimpleNamespace(data=data), dynamic_width=dynamic)
        fake._collate = lambda mb, fake=fake: collate(fake, mb)
        wlist = widths if dynamic else [800] * n
        tensors = [torch.full((1, 1, w), float(i)) for i, w in enumerate(wlist)]  # value = crop index
        order = np.argsort(wlist, kind="stable").tolist() if (k % 2 == 0 and n > 1) else None
        out = mk(fake, tensors, order)
        batch_cases.append({"widths": wlist, "dynamic": dynamic, "batch_size": data.batch_size, "width_budget": budget,
                            "max_batch_size": max_bs, "order": order,
                            "batches": [{"members": [int(v) for v in b[:, 0, 0, 0].tolist()], "width": int(b.shape[-1])} for b in out]})
    tok = ref_import("yomitoku.postprocessor.parseq_tokenizer")
    charset = [chr(0x3041 + i) for i in range(60)]
    t = tok.ParseqTokenizer(charset)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn(24, 12, len(t) - 2, generator=g) * 3.0
    logits[:, :, 0] += torch.linspace(-4, 6, 12)[None, :]  # <eos> grows more likely along the sequence
    logits[3, :, 0] = -30.0  # a row that never ends
    probs = logits.softmax(-1)
    texts, scores = t.decode(probs)
    tok_case = {"charset": charset, "probs": probs.tolist(), "texts": texts, "scores": scores}
    with open(os.path.join(GOLDEN, "geometry.json"), "w") as f:
        json.dump({"resize": resize_cases, "pad": pad_cases, "quads": quad_cases, "levels": level_cases, "batches": batch_cases,
                   "tokenizer": tok_case}, f)
    print(f"[geometry] resize {len(resize_cases)}, pad {len(pad_cases)}, quads {len(quad_cases)} "
          f"({sum(c['valid'] for c in quad_cases)} valid), levels {len(level_cases)}, batching {len(batch_cases)}, "
          f"tokenizer {len(texts)} rows (lengths {sorted(set(len(x) for x in texts))})")


def pin_configs():
    """Default values of every on-path config dataclass of the reference (configs/*.py), as plain containers ->
    tests/golden/configs.json.  Paths into the reference's resource directory are reduced to their base names."""
    import dataclasses
    import importlib.util
    import json

    from ._refstubs import REF_SRC

    names = {
        "TextDetectorDBNetConfig": "cfg_text_detector_dbnet", "TextDetectorDBNetV2Config": "cfg_text_detector_dbnet_v2",
        "TextDetectorDBNetV2_1Config": "cfg_text_detector_dbnet_v2_1", "TextRecognizerPARSeqConfig": "cfg_text_recognizer_parseq",
        "TextRecognizerPARSeqV2Config": "cfg_text_recognizer_parseq_v2", "TextRecognizerPARSeqSmallConfig": "cfg_text_recognizer_parseq_small",
        "TextRecognizerPARSeqTinyConfig": "cfg_text_recognizer_parseq_tiny",
        "TextRecognizerPARSeqLargeV41Config": "cfg_text_recognizer_parseq_large_v4_1",
        "TextRecognizerPARSeqTinyDynwV4Config": "cfg_text_recognizer_parseq_tiny_dynw_v4",
        "LayoutParserRTDETRv2Config": "cfg_layout_parser_rtdtrv2", "LayoutParserRTDETRv2V2Config": "cfg_layout_parser_rtdtrv2_v2",
        "TableStructureRecognizerRTDETRv2Config": "cfg_table_structure_recognizer_rtdtrv2",
    }

    def clean(v):
        if isinstance(v, dict):
            return {k: clean(x) for k, x in v.items()}
        if isinstance(v, (list, tuple)):
            return [clean(x) for x in v]
        if isinstance(v, str) and ("/" in v or "\\" in v) and os.path.splitext(v)[1] in (".txt", ".ttf", ".otf"):
            return "<resource>/" + os.path.basename(v)
        return v

    install = ref_import  # registers the namespace packages
    install("yomitoku.constants") if os.path.exists(os.path.join(REF_SRC, "yomitoku", "constants.py")) else None
    out = {}
    for cls, mod in names.items():
        path = os.path.join(REF_SRC, "yomitoku", "configs", mod + ".py")
        spec = importlib.util.spec_from_file_location("yomitoku.configs." + mod, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = m
        spec.loader.exec_module(m)
        out[cls] = clean(dataclasses.asdict(getattr(m, cls)()))
    with open(os.path.join(GOLDEN, "configs.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(f"[configs] wrote {len(out)} default configs")


def main(argv):
    what = argv[1] if len(argv) > 1 else "all"
    os.makedirs(GOLDEN, exist_ok=True)
    todo = {"dbnet": pin_dbnet, "parseq": pin_parseq, "rtdetr": pin_rtdetr, "host": pin_host_logic, "aggregate": pin_aggregate,
            "filters": pin_filters, "geometry": pin_geometry, "cells": pin_cells, "export": pin_export,
            "configs": pin_configs, "searchable_pdf": pin_searchable_pdf, "unit": pin_reference_unit_tests}
    for k, fn in todo.items():
        if what in (k, "all"):
            fn()


if __name__ == "__main__":
    main(sys.argv)

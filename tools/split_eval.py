"""How far do the nets' outputs move when the convolutions run with split operands (ymk_conv_split.hip) instead of
exact fp32 MFMA?  The question BASELINE.json's tolerance asks: probability maps and logits within 1e-3 of the fp32 CPU
path, discrete outputs unchanged.  For conv_split in SPLITS (env, default 0,3,2,16; 16 = two scaled fp16 planes): DBNet at the full page size (vs the oracle and vs the fp32
kernel), the reference-class goldens of PARSeq / RT-DETR, a wave-sized grouped PARSeq forward, an RT-DETR batch, and a
whole DocumentAnalyzer page (strings / boxes / order compared field by field).  Prints one JSON document."""
import ast
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from yomitoku_amd import _lib  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
dev = torch.device("cuda:0")


def rtdetr_match(lg, bx, ref_lg, ref_bx, tol_logit=1e-3, tol_box=1e-4):
    """Largest error in units of the tolerance after the one-to-one row matching of tests/test_rtdetr_gpu.py, and the
    fraction of rows matched at rank distance 0."""
    from scipy.optimize import linear_sum_assignment

    worst, same = 0.0, []
    for b in range(lg.shape[0]):
        d = np.maximum(np.abs(lg[b][:, None, :] - ref_lg[b][None, :, :]).max(-1) / tol_logit,
                       np.abs(bx[b][:, None, :] - ref_bx[b][None, :, :]).max(-1) / tol_box)
        gap = np.abs(np.arange(d.shape[0])[:, None] - np.arange(d.shape[1])[None, :])
        rows, match = linear_sum_assignment(np.where(gap <= 2, np.minimum(d, 1e3), 1e6))
        worst = max(worst, float(d[rows, match].max()))
        same.append(float((rows == match).mean()))
    return worst, float(np.mean(same))


def schema_diff(a, b, stats):
    """Walk two model_dump() trees: count differing discrete leaves, track the largest relative float difference."""
    if isinstance(a, dict):
        if a.keys() != b.keys():
            stats["discrete"] += 1
            return
        for k in a:
            schema_diff(a[k], b[k], stats)
    elif isinstance(a, (list, tuple)):
        if len(a) != len(b):
            stats["discrete"] += 1
            return
        for x, y in zip(a, b):
            schema_diff(x, y, stats)
    elif isinstance(a, float):
        stats["float_rel"] = max(stats["float_rel"], abs(a - b) / max(abs(a), abs(b), 1e-12))
    else:
        stats["leaves"] += 1
        if a != b:
            stats["discrete"] += 1


def main():
    from oracle.dbnet import dbnet_forward
    from oracle.preprocess import detector_preprocess
    from yomitoku_amd import DocumentAnalyzer, imaging
    from yomitoku_amd.nets import DBNet, RTDETRv2
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_line_batch, synthetic_page, synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_parseq_gpu import _net as parseq_net

    out = {}
    # ---- fixed inputs
    det_sd = dbnet_state_dict(1234)
    img = synthetic_page(21, 1600, 1200)
    det_ref = dbnet_forward(det_sd, detector_preprocess(img))["binary"]  # the fp32 CPU path (oracle)
    zg = np.load(os.path.join(GOLD, "dbnet_ref_64x96.npz"))
    rec_sd = parseq_state_dict(1235, eos_bias=5.5)
    groups = [synthetic_line_batch(100 + i, b, w) for i, (b, w) in enumerate([(40, 160), (10, 800), (64, 72), (30, 240), (20, 400), (48, 96), (24, 320), (12, 640)])]
    rt_sd = rtdetr_state_dict(1242, num_classes=6)
    rt_x = torch.rand(8, 3, 640, 640, generator=torch.Generator().manual_seed(9))
    pages = [synthetic_page_with_truth(3 + i, h, w)[0] for i, (h, w) in enumerate([(1600, 1200), (1200, 1600), (1600, 1200), (1000, 1400)])]
    lite = {"ocr": {"text_detector": {"from_pretrained": False},
                    "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True, "batch_bucketing": True, "source_downscale": True}},
            "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}}}
    base = {}
    for split in [int(v) for v in os.environ.get("SPLITS", "0,3,2,16").split(",")]:
        _lib.debug_option("conv_split", split)
        r = {}
        # DBNet
        net = DBNet().load_state_dict(det_sd).to(dev)
        x = imaging.detector_tensor(imaging.page_to_device(img, dev), 1280, 1600)
        p = net(x)["binary"].cpu()
        r["dbnet_full_page_max_dP_vs_oracle"] = float((p - det_ref).abs().max())
        gnet = DBNet().load_state_dict(dbnet_state_dict(int(zg["seed"]))).to(dev)
        r["dbnet_golden64x96_max_dP_vs_reference"] = float(np.abs(gnet(torch.from_numpy(zg["x"]).to(dev))["binary"].cpu().numpy() - zg["prob"]).max())
        xb = x.repeat(4, 1, 1, 1)
        pb = net(xb)["binary"].cpu()
        # PARSeq goldens (reference class)
        for tag in ("eos", "rep"):
            z = np.load(os.path.join(GOLD, f"parseq_ref_{tag}.npz"))
            _, pnet = parseq_net(dev, parseq_state_dict(**ast.literal_eval(str(z["ckpt"]))))
            lg = pnet(torch.from_numpy(z["x"]).to(dev)).cpu()
            r[f"parseq_golden_{tag}"] = {"steps_equal": bool(pnet.last_ar_steps == int(z["steps"])),
                                         "ids_equal": bool(lg.shape[:2] == z["ids"].shape and np.array_equal(lg.argmax(-1).numpy().astype(np.int32), z["ids"])),
                                         "max_d_top_logit": float(np.abs(lg.max(-1).values.numpy() - z["top"]).max()) if lg.shape[:2] == z["ids"].shape else None}
        # wave-sized grouped forward
        _, pnet = parseq_net(dev, rec_sd)
        logits, out_lens, steps = pnet.forward_groups([g.to(dev) for g in groups])
        lgw = logits.cpu()
        # RT-DETR
        rnet = RTDETRv2({"RTDETRTransformerv2": {"num_classes": 6, "num_queries": 300, "num_layers": 6, "hidden_dim": 256, "eval_spatial_size": [640, 640]}}).load_state_dict(rt_sd).to(dev)
        ro = rnet(rt_x.to(dev))
        rl, rb = ro["pred_logits"].cpu().numpy(), ro["pred_boxes"].cpu().numpy()
        for tag in ("layout", "table", "cell"):
            z = np.load(os.path.join(GOLD, f"rtdetr_ref_{tag}.npz"))
            seed, nc, size, nq = int(z["seed"]), int(z["num_classes"]), int(z["size"]), int(z["num_queries"])
            sd = rtdetr_state_dict(seed, num_classes=nc, eval_size=(size, size), enc_score_gain=1.0 if size == 640 else 12.0)
            gnet2 = RTDETRv2({"RTDETRTransformerv2": {"num_classes": nc, "num_queries": nq, "num_layers": 6, "hidden_dim": 256, "eval_spatial_size": [size, size]}}).load_state_dict(sd).to(dev)
            xg = torch.rand(1, 3, size, size, generator=torch.Generator().manual_seed(int(z["x_seed"])))
            o = gnet2(xg.to(dev))
            worst, same = rtdetr_match(o["pred_logits"].cpu().numpy(), o["pred_boxes"].cpu().numpy(), z["logits"], z["boxes"])
            r[f"rtdetr_golden_{tag}"] = {"worst_error_in_tolerances": round(worst, 4), "rows_at_rank_distance_0": round(same, 4)}
        # whole pages
        an = DocumentAnalyzer(configs=lite, device="cuda:0")
        an.text_detector.model.load_state_dict(dbnet_state_dict(1234, out_bias=-2.0))
        an.text_recognizer.model.load_state_dict(parseq_state_dict(1235, eos_bias=6.0))
        an.layout.layout_parser.model.load_state_dict(rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0))
        an.layout.table_structure_recognizer.model.load_state_dict(rtdetr_state_dict(1241, num_classes=3, score_bias=-1.0))
        dumps = [res.model_dump() for res in an.serve(pages, wave=4, in_flight=2)]
        an.close()
        if split == 0:
            base = dict(p=p, pb=pb, lgw=lgw, out_lens=out_lens, steps=steps, rl=rl, rb=rb, dumps=dumps)
        else:
            r["dbnet_full_page_max_dP_vs_fp32_kernel"] = float((p - base["p"]).abs().max())
            r["dbnet_batch4_max_dP_vs_fp32_kernel"] = float((pb - base["pb"]).abs().max())
            same_steps = list(steps) == list(base["steps"])
            row, worst, arg_same, n = 0, 0.0, 0, 0
            for g, n_out in zip(groups, base["out_lens"]):
                a, b = lgw[row : row + g.shape[0], :n_out], base["lgw"][row : row + g.shape[0], :n_out]
                worst = max(worst, float((a - b).abs().max()))
                arg_same += int((a.argmax(-1) == b.argmax(-1)).sum())
                n += a.shape[0] * a.shape[1]
                row += g.shape[0]
            r["parseq_wave_vs_fp32_kernel"] = {"lines": int(lgw.shape[0]), "ar_steps_equal": same_steps, "max_d_logit": worst, "argmax_equal_frac": arg_same / n}
            worst, same = rtdetr_match(rl, rb, base["rl"], base["rb"])
            r["rtdetr_batch8_vs_fp32_kernel"] = {"worst_error_in_tolerances": round(worst, 4), "rows_at_rank_distance_0": round(same, 4)}
            st = {"discrete": 0, "leaves": 0, "float_rel": 0.0}
            for a, b in zip(dumps, base["dumps"]):
                schema_diff(a, b, st)
            r["pages_vs_fp32_kernel"] = {"pages": len(dumps), "words": sum(len(d["words"]) for d in dumps), "discrete_leaves": st["leaves"],
                                         "discrete_leaves_differing": st["discrete"], "max_rel_score_diff": st["float_rel"]}
        out[f"conv_split={split}"] = r
    _lib.debug_option("conv_split", 0)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

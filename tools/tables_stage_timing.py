"""The table-structure stage of a wave, alone on the device and taken apart: for the 16 pages of a wave of bench.py's
table-heavy leg (the calibrated heads' own ~10 table boxes per page) - the host part of the crop pre-processing, the
launch of the crops, the forward's launch time, the wait for its results - per forward and per wave, with a device
synchronisation between the parts so that each is seen alone.

    python tools/tables_stage_timing.py [--waves 3] [--max-tables 64]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--waves", type=int, default=3)
    ap.add_argument("--wave", type=int, default=16)
    ap.add_argument("--max-tables", type=int, default=0)
    args = ap.parse_args()
    from yomitoku_amd import DocumentAnalyzer, imaging

    device = bench.rank_device(0)
    sds = bench.calibrate_heads(bench.make_checkpoints("lite"), device, bench.Page(0, device))
    pages = bench.make_pages(list(range(args.wave * args.waves)), device)
    an = DocumentAnalyzer(configs=bench.MODEL_SETS["lite"], device=str(device))
    for net, key in zip(bench.analyzer_nets(an), ("det", "rec", "lay", "tab")):
        net.load_state_dict(sds[key])
    lp, ts = an.layout.layout_parser, an.layout.table_structure_recognizer
    if args.max_tables:
        ts.MAX_TABLES_PER_FORWARD = args.max_tables
    rows = []
    for w in range(args.waves + 1):  # wave 0 warms up
        chunk = pages[(w % args.waves) * args.wave : (w % args.waves + 1) * args.wave]
        devs = [p.dev for p in chunk]
        boxes = [[t.box for t in l.tables] for l in lp.pages_from_raw(lp.forward_pages(devs))]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        raw = ts.forward_tables(devs, boxes)
        torch.cuda.synchronize()
        whole = time.perf_counter() - t0
        # the same forwards, part by part
        flat = [(p, b) for p, bs in enumerate(boxes) for b in bs]
        n_fwd = max(1, -(-len(flat) // ts.MAX_TABLES_PER_FORWARD))
        per = -(-len(flat) // n_fwd) if flat else 1
        parts = {"host_tables_ms": 0.0, "crops_launch_ms": 0.0, "crops_device_ms": 0.0, "forward_launch_ms": 0.0, "forward_device_ms": 0.0, "d2h_ms": 0.0}
        for start in range(0, len(flat), per):
            sub = flat[start : start + per]
            t = time.perf_counter()
            for p, b in sub:  # the coefficient tables alone (what rtdetr_batch_tensor computes on the host)
                x1, y1, x2, y2 = imaging._clamped_box(devs[p], b)
                imaging.pil_bilinear_coeffs(x2 - x1, 640)
                imaging.pil_bilinear_coeffs(y2 - y1, 640)
            parts["host_tables_ms"] += (time.perf_counter() - t) * 1e3
            t = time.perf_counter()
            batch, metas = imaging.rtdetr_batch_tensor(devs, sub, (640, 640))
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            preds = ts.model(batch)
            t3 = time.perf_counter()
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            preds["pred_logits"].cpu().numpy(), preds["pred_boxes"].cpu().numpy()
            t5 = time.perf_counter()
            parts["crops_launch_ms"] += (t1 - t) * 1e3
            parts["crops_device_ms"] += (t2 - t1) * 1e3
            parts["forward_launch_ms"] += (t3 - t2) * 1e3
            parts["forward_device_ms"] += (t4 - t3) * 1e3
            parts["d2h_ms"] += (t5 - t4) * 1e3
        if w:
            rows.append(dict({"tables": len(flat), "forwards": n_fwd, "forward_tables_ms": round(whole * 1e3, 2)}, **{k: round(v, 2) for k, v in parts.items()}))
    out = {"max_tables_per_forward": ts.MAX_TABLES_PER_FORWARD, "waves": rows,
           "mean": {k: round(float(np.mean([r[k] for r in rows])), 2) for k in rows[0]}}
    print(json.dumps(out))
    an.close()


if __name__ == "__main__":
    main()

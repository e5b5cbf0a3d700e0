"""One wave of pages through the truth-driven analyzer of bench.py, stage by stage (wall time with a device sync after
each stage), then a cProfile of a whole wave (host hot spots) and - with YMK_PROF_DUMP=1 - the per-launch conv table."""
import cProfile
import ctypes
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from yomitoku_amd import _lib

dev = torch.device("cuda:0")
W = int(os.environ.get("W", 8))
SET = os.environ.get("SET", "lite")
sds = bench.make_checkpoints(SET)
pages = bench.make_pages(range(1000, 1000 + 2 * W), dev)
sds = bench.calibrate_heads(sds, dev, bench.Page(0, dev))
an = bench.build_analyzer(dev, sds, SET)
an.truth = pages
an.concurrent_chains = False
wave = pages[:W]


def run(ws):
    return an.analyze_pages([p.dev for p in ws], wave=len(ws))  # page ids 0.. line up with an.truth for the first wave


run(wave)
torch.cuda.synchronize()


def T(label, f, *a):
    torch.cuda.synchronize()
    t = time.perf_counter()
    r = f(*a)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{label:28s} host {1e3 * (t1 - t):8.2f} ms   +drain {1e3 * (t2 - t1):7.2f} ms   per page {1e3 * (t2 - t) / W:6.2f} ms", flush=True)
    return r


devs = [p.dev for p in wave]
det, rec = an.text_detector, an.text_recognizer
maps = T("det.forward_pages (+D2H)", det.forward_pages, devs)
sizes = [tuple(int(v) for v in p.shape[:2]) for p in devs]
T("det.extract_boxes(truth)", det.extract_boxes, [p.truth_map for p in wave], sizes)
preps = T("rec.preprocess x W", lambda: [rec.preprocess(d, p.quads) for d, p in zip(devs, wave)])
jobs = [(ds, plans) for batches, _, ds, _ in preps for plans in batches]
print("mini-batches", len(jobs), "lines", sum(len(j[1]) for j in jobs))
tensors = T("rec.collate all", lambda: [rec._collate(ds, plans) for ds, plans in jobs])
out = T("rec.forward_groups", rec.model.forward_groups, tensors)
print("   ar steps per group: max", max(out[2]), "mean", sum(out[2]) / len(out[2]))
st = T("rec.token_stats+cpu", lambda: [t.cpu().numpy() for t in rec.model.token_stats(out[0])])
T("rec.recognize_pages total", rec.recognize_pages, devs, [p.quads for p in wave])
lp, ts = an.layout.layout_parser, an.layout.table_structure_recognizer
T("layout.parse_pages", lp.parse_pages, devs)
T("tables.recognize_pages", ts.recognize_pages, devs, [p.tables for p in wave])
from yomitoku_amd import imaging as _im
flat = [(d, b) for d, p in zip(devs, wave) for b in p.tables]
xb = torch.empty((len(flat), 3, 640, 640), dtype=torch.float32, device=dev)
metas = T("  tables: resize crops", lambda: [_im.rtdetr_tensor(d, b, (640, 640), out=xb[k])[1:] for k, (d, b) in enumerate(flat)])
preds = T("  tables: forward", ts.model, xb)
lg, bx = T("  tables: D2H", lambda: (preds["pred_logits"].cpu().numpy(), preds["pred_boxes"].cpu().numpy()))
T("  tables: postprocess", lambda: [ts.postprocess({"pred_logits": lg[k:k+1], "pred_boxes": bx[k:k+1]}, {"size": m[0], "offset": m[1]}) for k, m in enumerate(metas)])
print("tables in wave", sum(len(p.tables) for p in wave))
from yomitoku_amd.serving import Wave

T("_ocr_wave", an._ocr_wave, Wave(0, range(W), devs, devs))
T("_layout_wave", an._layout_wave, Wave(0, range(W), devs, devs))
T("wave total (serial chains)", run, wave)
an.concurrent_chains = True
T("wave total (2 streams)", run, wave)
an.concurrent_chains = False

pr = cProfile.Profile()
pr.enable()
run(wave)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

if os.environ.get("YMK_PROF_DUMP"):
    lib = _lib.load()
    _lib.debug_option("prof_dump", 1)
    lib.ymk_prof_begin()
    run(wave)
    torch.cuda.synchronize()
    ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln))
    print("conv total ms", ms.value, "GF", fl.value / 1e9, "launches", ln.value)

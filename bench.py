#!/usr/bin/env python
"""Benchmark of the MI355X DocumentAnalyzer hot path (contract: task prompt / DESIGN.md §8).

    python bench.py --gpus N --steps K --warmup W [--workload analyzer|detector|recognizer] [--pages 64] [--procs 4] [--workers 2]

One rank per GPU (torchrun env).  A "step" is one pass of the hot path over one batch of synthetic
1600x1200 pages that are already resident in HBM (uint8 BGR, as `cv2.imread` would hand them over).
Rank 0 prints ONE JSON line: BASELINE.json's metric (pages/s, whole job), the roofline of the
dominant kernel (live HIP-event timing of every implicit-GEMM convolution launch on its own stream)
and, at N=1, the CPU baseline (oracle restatement of the reference's PyTorch-CPU path, bounded sample).

workload analyzer (default; BASELINE.json configs[3], `--lite` model set): DBNet text detector,
  PARSeq tiny-dynw recogniser (dynamic_width + batch_bucketing), RT-DETRv2 layout parser, RT-DETRv2
  table-structure recogniser, host post-processing and aggregation.  Every page goes through
  `DocumentAnalyzer.__call__` on its own, as in the reference (cli/main.py:116-120); `--procs` processes per
  GPU x `--workers` pages in flight each (yomitoku_amd/parallel.py: the host half of a page is Python, one GIL
  per process).  Weights are seeded random draws (no
  network) which detect noise, so the DISCRETE hand-overs between stages use the page generator's
  ground truth - the recogniser gets the true text-line quads, the table recogniser the true table
  boxes, the aggregation the true paragraph boxes - while every network and every pre/post stage
  still runs at full cost on every page inside the timed region.
workload detector (configs[1]): DBNet forward alone on a batch of 8 pages.
workload recognizer (configs[2]): TextRecognizer (`--rec-model parseq` = open-beta geometry, or the lite model) on 2048
  synthetic text lines per step; metric text-lines/sec.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD

# YMK_BENCH_DRY=1: CPU rehearsal of the ORCHESTRATION only (ranks over gloo, helper processes, step barriers, the
# max-over-ranks clock, teardown) with stub page workers from tests/bench_dry_stubs.py - what tests/test_bench_dry.py
# runs at world size 2, since multi-GPU boxes are the driver's.  It does no GPU work and its line says "dry_run".
DRY = os.environ.get("YMK_BENCH_DRY") == "1"


def rank_device(local_rank):
    if DRY:
        return torch.device("cpu")
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    return device


def device_sync():
    if not DRY:
        torch.cuda.synchronize()

LITE_CONFIGS = {
    "ocr": {
        "text_detector": {"from_pretrained": False},
        "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                            "batch_bucketing": True, "source_downscale": True},  # cli/main.py:505-520 (--lite)
    },
    "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}},
}
# Biases of the seeded heads, chosen so that the random nets emit realistic unit COUNTS: sparse blobs
# above the DB threshold, text of ~10-30 tokens per line batch, a handful of layout / row / column boxes.
CKPT = dict(det=dict(seed=1234), rec=dict(seed=1235, eos_bias=6.1), lay=dict(seed=1240, num_classes=6, score_gain=3.0),
            tab=dict(seed=1241, num_classes=3, score_gain=3.0))


if os.environ.get("YMK_BENCH_CKPT"):  # e.g. '{"rec": {"eos_bias": 5.5}}' - for unit-count sweeps
    for _k, _v in json.loads(os.environ["YMK_BENCH_CKPT"]).items():
        CKPT[_k].update(_v)


def make_checkpoints():
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    return {"det": dbnet_state_dict(**CKPT["det"]), "rec": parseq_state_dict(**CKPT["rec"]),
            "lay": rtdetr_state_dict(**CKPT["lay"]), "tab": rtdetr_state_dict(**CKPT["tab"])}


def calibrate_heads(sds, device, page):
    """Random heads either fire on hundreds of queries or on none (their logits are tightly clustered), so
    the last bias of each head is shifted - once, on rank 0, from the net's own logits on the first page -
    until a realistic number of units clears the module's threshold: ~16 layout boxes (> 0.5), ~12 table
    rows/columns (> 0.4), ~2 % of the detector map above the DB threshold (0.3).  Deterministic."""
    import math

    from yomitoku_amd import imaging
    from yomitoku_amd.nets import DBNet, RTDETRv2

    def shift_for(logits, target, thresh):
        v = torch.sort(logits.flatten(), descending=True).values
        cut = 0.5 * (v[target - 1] + v[target]).item()
        return math.log(thresh / (1 - thresh)) - cut

    for key, nc, target, thresh, box in (("lay", 6, 16, 0.5, None), ("tab", 3, 12, 0.4, page.tables[0] if page.tables else None)):
        net = RTDETRv2({"RTDETRTransformerv2": {"num_classes": nc}}).load_state_dict(sds[key]).to(device)
        x, _, _ = imaging.rtdetr_tensor(page.dev, box)
        lg = net(x[None])["pred_logits"].float().cpu()
        bias = sds[key]["decoder.dec_score_head.5.bias"].clone()
        if key == "tab":  # per class: ~5 rows, ~5 columns, ~2 spans, so that the cell grid is not empty
            for c, tgt in enumerate((5, 5, 2)):
                bias[c] += shift_for(lg[..., c], tgt, thresh)
        else:
            bias += shift_for(lg, target, thresh)
        sds[key]["decoder.dec_score_head.5.bias"] = bias
        net.close()
    net = DBNet().load_state_dict(sds["det"]).to(device)
    p = net(imaging.detector_tensor(page.dev, 1280, 1600))["binary"].float().cpu().flatten()
    logit = torch.log(p.clamp(1e-7, 1 - 1e-7) / (1 - p.clamp(1e-7, 1 - 1e-7)))
    k = int(0.02 * logit.numel())
    cut = torch.topk(logit, k).values[-1].item()
    sds["det"]["decoder.binarize.6.bias"] = sds["det"]["decoder.binarize.6.bias"] + (math.log(0.3 / 0.7) - cut)
    net.close()
    return sds


class Page:
    def __init__(self, seed, device):
        from yomitoku_amd import imaging
        from yomitoku_amd.utils.synth import synthetic_page_with_truth

        self.img, self.quads, self.tables, self.paragraphs = synthetic_page_with_truth(seed)
        self.dev = imaging.page_to_device(self.img, device)


def build_analyzer(device, sds):
    import logging

    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd.schemas import Element, LayoutAnalyzerSchema, TextDetectorSchema

    logging.getLogger("yomitoku_amd.base").setLevel(logging.WARNING)

    class TruthDrivenAnalyzer(da.DocumentAnalyzer):
        """DocumentAnalyzer whose stage hand-overs use the page's ground truth (see module doc)."""

        truth = None
        stats = None

        def _detect_and_recognize(self, page):
            det_noise, _ = self.text_detector(page)  # full detector stage; its noise boxes are not propagated
            det = TextDetectorSchema(points=self.truth.quads, scores=[1.0] * len(self.truth.quads))
            rec, ocr = self.text_recognizer(page, det.points, None)
            if self.stats is not None:
                self.stats["det_boxes"].append(len(det_noise.points))
            return det, rec, ocr

    class TruthLayout:
        def __init__(self, inner, owner):
            self.inner, self.owner = inner, owner

        def __call__(self, page):
            noise, _ = self.inner.layout_parser(page)  # full layout stage, result not propagated
            tables, _ = self.inner.table_structure_recognizer(page, self.owner.truth.tables)
            paragraphs = [Element(id=None, box=b, score=1.0, role=None, contents=None) for b in self.owner.truth.paragraphs]
            if self.owner.stats is not None:
                self.owner.stats["layout_boxes"].append(len(noise.paragraphs) + len(noise.tables) + len(noise.figures))
                self.owner.stats["cells"].append(sum(len(t.cells) for t in tables))
            return LayoutAnalyzerSchema(paragraphs=paragraphs, tables=tables, figures=[]), None

    an = TruthDrivenAnalyzer(configs=LITE_CONFIGS, device=str(device))
    an.text_detector.model.load_state_dict(sds["det"])
    an.text_recognizer.model.load_state_dict(sds["rec"])
    an.layout.layout_parser.model.load_state_dict(sds["lay"])
    an.layout.table_structure_recognizer.model.load_state_dict(sds["tab"])
    an.layout = TruthLayout(an.layout, an)

    def run(page: Page):
        an.truth = page
        return an(page.dev)[0]

    run.analyzer = an
    return run


def cpu_analyzer_page(sds, page: Page, charset):
    """The same page through the oracle chain on the host cores (what `--lite -d cpu` computes)."""
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg

    op.detect(sds["det"], page.img)
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    op.recognize(sds["rec"], ocfg, page.img, page.quads, charset, dynamic_width=True, batch_bucketing=True,
                 width_budget=8000, max_batch_size=64, batch_size=10, source_downscale=True)
    op.layout(sds["lay"], page.img)
    op.tables(sds["tab"], page.img, page.tables)


def _helper_init(local_rank, sds, shares, workers, index):
    """One helper process of a rank (yomitoku_amd/parallel.py PageProcesses): same GPU, own HIP context and
    analyzer replicas built from the rank's checkpoints (received as numpy arrays), own pages resident in HBM."""
    from yomitoku_amd.parallel import PageParallel

    sds = {k: {name: torch.from_numpy(a) for name, a in sd.items()} for k, sd in sds.items()}
    device = rank_device(local_rank)
    pages = make_pages(shares[index], device)
    pool = PageParallel(lambda i: build_analyzer(device, sds), n_workers=workers)
    pool.map(pages[:workers])  # first-call allocations happen before the parent starts its clock
    device_sync()

    def step(_payload):
        pool.map(pages)
        device_sync()
        return len(pages)

    return step


def make_pages(seeds, device):
    from concurrent.futures import ThreadPoolExecutor as _TPE

    with _TPE(max_workers=8) as ex:  # numpy's generators release the GIL: 64 pages in ~2 s instead of ~11 s
        return list(ex.map(lambda i: Page(i, device), seeds))


REC_PRESETS = {  # --rec-model: (synth.parseq_state_dict kwargs, oracle preset name, batching kwargs of the oracle chain)
    "parseq": (dict(seed=1236, patch=(8, 8), enc_dim=512, dec_dim=512, num_tokens=7312, eos_bias=6.5), "parseq",
               dict(width_budget=None, max_batch_size=None, batch_size=128)),
    "parseq-tiny-dynw-v4": (dict(seed=1235, eos_bias=6.1), "parseq-tiny-dynw-v4",
                            dict(width_budget=8000, max_batch_size=64, batch_size=10)),
}


def recognizer_workload(args, rank, local_rank, world, device, lib):
    """BASELINE.json configs[2]: TextRecognizer with dynamic_width + batch_bucketing on 2048 synthetic 32 x W text-line
    crops per GPU per step (one sheet image + 2048 quads through TextRecognizer.__call__).  Unit: text lines."""
    from yomitoku_amd import _lib, imaging
    from yomitoku_amd import distributed as ydist
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_sheet

    ckpt_kw, preset, batching = REC_PRESETS[args.rec_model]
    sd = ydist.broadcast_state_dict(parseq_state_dict(**ckpt_kw) if rank == 0 else None, src=0, device=device)
    sheet, quads = synthetic_line_sheet(seed=1 + rank, n_lines=args.lines)
    page = imaging.page_to_device(sheet, device)
    rec = TextRecognizer(model_name=args.rec_model, from_pretrained=False, device=str(device), dynamic_width=True, batch_bucketing=True,
                         num_parallel_batches=args.rec_lanes)
    rec.model.load_state_dict(sd)

    def step():
        return rec(page, quads)[0]

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    if rank != 0:
        return None
    roof = cpu = None
    rec.num_parallel_batches = 1  # the per-launch event bookkeeping is single-threaded: serial pass for the roofline leg
    _lib.check(lib.ymk_prof_begin())
    step()
    torch.cuda.synchronize()
    ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
    _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
    if ms.value > 0:
        ach = fl.value / (ms.value * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": "conv_igemm / conv_splitk (every linear layer of the ViT encoder and the decoder)",
                "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": None, "launches_per_line": round(ln.value / args.lines, 2),
                "gflop_per_line": round(fl.value / args.lines / 1e9, 2), "kernel_ms_per_step": round(ms.value, 2)}
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pipeline as op
        from oracle.parseq import PRESETS, make_cfg

        n_cpu = min(args.lines, 96)
        ocfg = make_cfg(**PRESETS[preset])
        t1 = time.perf_counter()
        op.recognize(sd, ocfg, sheet, quads[:n_cpu], rec.charset, dynamic_width=True, batch_bucketing=True, **batching)
        cpu = {"value": round(n_cpu / (time.perf_counter() - t1), 2), "unit": "lines/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"the first {n_cpu} of the same lines through oracle.pipeline.recognize (PyTorch-CPU fp32 restatement)"}
    widths = [q[1][0] - q[0][0] for q in quads]
    return {
        "metric": f"PARSeq text-lines/sec (TextRecognizer {args.rec_model}, dynamic_width + batch_bucketing)",
        "value": round(args.lines * args.steps * world / dt, 1), "unit": "lines/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"TextRecognizer {args.rec_model} on {args.lines} synthetic 32 x W text lines per GPU per step "
                               f"(W log-normal, median {int(np.median(widths))} px, [16, 800]; BASELINE.json configs[2]); one "
                               f"TextRecognizer call per step, {len(set(out.contents))} distinct strings decoded, "
                               f"mean length {np.mean([len(c) for c in out.contents]):.1f} characters (seeded random weights)",
                   "lines_per_step_per_gpu": args.lines,
                   "parallelism": f"line sheets sharded x{world} GPU(s), {args.rec_lanes} mini-batches in flight",
                   "checkpoints": "seeded synthetic (no network)", "last_batch_ar_steps": int(rec.model.last_ar_steps)},
        "roofline": roof, "cpu_baseline": cpu,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="analyzer", choices=["analyzer", "detector", "recognizer"])
    ap.add_argument("--rec-model", default="parseq", choices=sorted(REC_PRESETS), help="recognizer workload: model (configs[2]: parseq)")
    ap.add_argument("--lines", type=int, default=2048, help="recognizer workload: text lines per step per GPU")
    ap.add_argument("--rec-lanes", type=int, default=4, help="recognizer workload: TextRecognizer num_parallel_batches")
    ap.add_argument("--pages", type=int, default=64, help="pages per step per GPU (BASELINE.json configs[3]: 64)")
    ap.add_argument("--procs", type=int, default=4, help="processes per GPU (analyzer workload): each has its own interpreter/GIL")
    ap.add_argument("--workers", type=int, default=2, help="pages in flight per process (analyzer workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from yomitoku_amd import _lib
    from yomitoku_amd import distributed as ydist
    from yomitoku_amd import imaging
    from yomitoku_amd.parallel import PageParallel

    rank, local_rank, world = ydist.init("gloo" if DRY else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert DRY or torch.cuda.is_available(), "bench.py needs a HIP device"
    device = rank_device(local_rank)
    lib = _lib.load()

    if args.workload == "recognizer":
        line = recognizer_workload(args, rank, local_rank, world, device, lib)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            torch.distributed.barrier()  # rank 0 is still in its roofline / CPU legs: the others wait here, not in teardown
            torch.distributed.destroy_process_group()
        return

    # ---- weights: drawn once on rank 0, ONE flat RCCL broadcast per checkpoint over xGMI
    sds = make_checkpoints() if rank == 0 else {k: None for k in ("det", "rec", "lay", "tab")}
    if rank == 0 and args.workload == "analyzer":
        sds = calibrate_heads(sds, device, Page(0, device))
    for k in ("det", "rec", "lay", "tab"):
        sds[k] = ydist.broadcast_state_dict(sds[k], src=0, device=device)

    # ---- synthetic pages of this rank, resident in HBM before the clock starts
    if args.workload == "detector":
        args.pages = min(args.pages, 8)
    seeds = [1000 * rank + i for i in range(args.pages)]
    n_procs = max(1, args.procs) if args.workload == "analyzer" else 1
    shares = [seeds[i::n_procs] for i in range(n_procs)]  # pages of this rank, dealt to its processes
    pages = make_pages(shares[0], device)
    extra = {}
    helpers = None
    if args.workload == "analyzer":
        from yomitoku_amd.parallel import PageProcesses

        if n_procs > 1:
            # the rank's checkpoints go to its helpers by value, as numpy arrays through the spawn pipe (no /dev/shm)
            wire = {k: {name: t.numpy() for name, t in sd.items()} for k, sd in sds.items()}
            helpers = PageProcesses(_helper_init, (local_rank, wire, shares, args.workers), n_procs=n_procs - 1, first_index=1)
        pool = PageParallel(lambda i: build_analyzer(device, sds), n_workers=args.workers)

        def step():  # every process of the rank walks its share of the rank's pages; the step ends when all have
            if helpers:
                helpers.start([None] * len(helpers))
            out = pool.map(pages)[-1]
            device_sync()
            if helpers:
                assert sum(helpers.finish()) + len(pages) == args.pages
            return out

        metric = "pages/sec (DocumentAnalyzer @1600x1200, lite model set)"
        workload = (f"Full DocumentAnalyzer per page (BASELINE.json configs[3]): DBNet dbnetv2_1 + PARSeq parseq-tiny-dynw-v4 "
                    f"(dynamic_width, batch_bucketing) + RT-DETRv2 layout + RT-DETRv2 table structure + host post-processing "
                    f"and aggregation; {args.pages} synthetic 1600x1200 pages per step per GPU, dealt to {n_procs} process(es) x "
                    f"{args.workers} pages in flight each; "
                    f"stage hand-overs use ground truth ({np.mean([len(p.quads) for p in pages]):.0f} text lines, "
                    f"{np.mean([len(p.tables) for p in pages]):.1f} tables, {np.mean([len(p.paragraphs) for p in pages]):.0f} "
                    f"paragraphs per page) because seeded random weights detect noise")
    else:
        from yomitoku_amd.nets import DBNet

        net = DBNet().load_state_dict(sds["det"]).to(device)
        x = torch.cat([imaging.detector_tensor(p.dev, 1280, 1600) for p in pages], 0)

        def step():
            return net(x)["binary"]

        metric = "pages/sec (TextDetector DBNet forward @1600x1200 -> 3x1600x1184)"
        workload = f"TextDetector DBNet forward alone, batch={args.pages} synthetic 1600x1200 pages per GPU (BASELINE.json configs[1])"

    for _ in range(args.warmup):
        step()
    device_sync()
    if world > 1:
        torch.distributed.barrier()
    device_sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    device_sync()
    if world > 1:
        torch.distributed.barrier()
    device_sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        dt = float(tt.item())
    assert out is not None

    if helpers:
        helpers.close()

    # ---- roofline leg: per-launch HIP events around the conv kernel.  The event bookkeeping is
    # single-threaded, so this pass walks the pages serially with ONE analyzer and ONE worker thread.
    roof = None
    if rank == 0 and not DRY:
        if args.workload == "analyzer":
            from concurrent.futures import ThreadPoolExecutor

            solo = pool.workers[0]
            solo.analyzer._pool.shutdown(wait=True)
            solo.analyzer._pool = ThreadPoolExecutor(max_workers=1)
            solo.analyzer.stats = {"det_boxes": [], "layout_boxes": [], "cells": []}
            prof_pages = pages[: min(8, len(pages))]

            def prof_step():
                for p in prof_pages:
                    solo(p)

            units = len(prof_pages)
        else:
            prof_step, units = step, args.pages
        _lib.check(lib.ymk_prof_begin())
        prof_step()
        torch.cuda.synchronize()
        ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
        alg_bytes = ctypes.c_double()
        _lib.check(lib.ymk_prof_bytes(ctypes.byref(alg_bytes)))
        if ms.value > 0:
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            roof = {
                "bound": "mfma",
                "kernel": "conv_igemm / conv_splitk (fp32 MFMA implicit GEMM: every conv / linear layer of the four nets)",
                "achieved": round(achieved, 2),
                "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": None,
                "algorithmic_bytes_per_launch": int(alg_bytes.value // max(1, ln.value)),
                "launches_per_page": int(ln.value // units),
                "avg_launch_us": round(ms.value * 1e3 / max(1, ln.value), 2),
                "kernel_ms_per_page": round(ms.value / units, 3),
                "gflop_per_page": round(fl.value / units / 1e9, 1),
            }
        pmc = os.path.join(ROOT, "profiles", f"r01_{args.workload}_pmc_conv_traffic.json")
        if roof is not None and os.path.exists(pmc):
            # HBM bytes per conv launch from the PMC passes of this same workload (rocprofv3 cannot run inside bench.py:
            # profiles/README.md has the commands); compare with algorithmic_bytes_per_launch
            with open(pmc) as f:
                t = json.load(f)
            roof["traffic"] = t["hbm_bytes_per_launch"]
            roof["traffic_source"] = os.path.relpath(pmc, ROOT)
        if args.workload == "analyzer" and roof is not None:
            # the north star quotes MFMA utilisation "on DBNet conv": the same measurement over the detector's launches alone
            det = solo.analyzer.text_detector
            _lib.check(lib.ymk_prof_begin())
            for p in prof_pages[:4]:
                det.model(det.preprocess(p.dev))
            torch.cuda.synchronize()
            _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
            d_ach = fl.value / (ms.value * 1e-3) / 1e12
            roof["dbnet_conv"] = {"achieved": round(d_ach, 2), "frac": round(d_ach / FP32_MFMA_PEAK_TFLOPS, 4),
                                  "launches_per_page": int(ln.value // 4), "kernel_ms_per_page": round(ms.value / 4, 3),
                                  "gflop_per_page": round(fl.value / 4 / 1e9, 1)}
        if args.workload == "analyzer" and not DRY:
            st = solo.analyzer.stats
            extra["measured_units_per_page"] = {
                "ar_steps_last_batch": int(solo.analyzer.text_recognizer.model.last_ar_steps),
                "noise_detector_boxes": float(np.mean(st["det_boxes"])),
                "noise_layout_boxes": float(np.mean(st["layout_boxes"])),
                "table_cells": float(np.mean(st["cells"])),
            }

    # ---- CPU baseline leg (rank 0, N=1): oracle chain on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRY:
        t1 = time.perf_counter()
        n_cpu = 0
        if args.workload == "analyzer":
            charset = pool.workers[0].analyzer.text_recognizer.charset
            while n_cpu < 1 or (time.perf_counter() - t1 < 12.0 and n_cpu < 3):
                cpu_analyzer_page(sds, pages[n_cpu % len(pages)], charset)
                n_cpu += 1
            sample = (f"{n_cpu} of the same synthetic pages through the oracle restatement (PyTorch-CPU fp32) of the `--lite` "
                      "chain: detector + recogniser + layout + table nets with their pre/post-processing, "
                      "PyTorch path for the detector too (onnxruntime is not installed)")
        else:
            from oracle.dbnet import dbnet_forward

            xc = x[:1].cpu()
            while n_cpu < 2 or (time.perf_counter() - t1 < 10.0 and n_cpu < 6):
                dbnet_forward(sds["det"], xc)
                n_cpu += 1
            sample = f"{n_cpu} pages of 1x3x1600x1184 through oracle/dbnet.py (detector net only)"
        cdt = time.perf_counter() - t1
        cpu = {"value": round(n_cpu / cdt, 4), "unit": "pages/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": sample}

    if rank == 0:
        total_pages = args.pages * args.steps * world
        line = {
            "metric": metric,
            "value": round(total_pages / dt, 3),
            "unit": "pages/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": workload, "pages_per_step_per_gpu": args.pages,
                       "parallelism": (f"page-sharded x{world} GPU(s), {n_procs} process(es) x {args.workers} pages in flight per GPU"
                                       if args.workload == "analyzer" else f"page-sharded x{world} GPU(s), one batch of {args.pages} per forward"),
                       "checkpoints": "seeded synthetic (no network)", **extra},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        if DRY:
            line["dry_run"] = True
            line["metric"] = "DRY RUN - orchestration rehearsal with stub page workers, not a measurement"
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()  # rank 0 is still in its roofline leg: the others wait here, not in teardown
        torch.distributed.destroy_process_group()


if DRY:  # stub page workers / pages / checkpoints for the CPU rehearsal (also in the spawned helper processes)
    from tests import bench_dry_stubs

    bench_dry_stubs.install(globals())

if __name__ == "__main__":
    main()

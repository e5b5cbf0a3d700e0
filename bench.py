#!/usr/bin/env python
"""Benchmark of the MI355X DocumentAnalyzer hot path (contract: task prompt / DESIGN.md §8).

    python bench.py --gpus N --steps K --warmup W [--workload analyzer|detector] [--pages 8]

One rank per GPU (torchrun env).  A "step" is one pass of the hot path over one batch of synthetic
1600x1200 pages that are already resident in HBM (uint8 BGR, as `cv2.imread` would hand them over).
Rank 0 prints ONE JSON line: BASELINE.json's metric (pages/s, whole job), the roofline of the
dominant kernel (live HIP-event timing of every implicit-GEMM convolution launch on its own stream)
and, at N=1, the CPU baseline (oracle restatement of the reference's PyTorch-CPU path, bounded sample).

workload analyzer (default; BASELINE.json configs[3], `--lite` model set): DBNet text detector,
  PARSeq tiny-dynw recogniser (dynamic_width + batch_bucketing), RT-DETRv2 layout parser, RT-DETRv2
  table-structure recogniser, host post-processing and aggregation, one page at a time like the
  reference (cli/main.py:116-120).  Weights are seeded random draws (no network), which detect noise,
  so the recogniser and table stages are driven with the page generator's ground-truth text-line
  quads / table boxes (~70 lines, 1-2 tables per page); every network and every pre/post stage still
  runs at full cost inside the timed region.
workload detector (configs[1]): DBNet forward alone on a batch of 8 pages.
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


def log(*a):
    print(*a, file=sys.stderr, flush=True)


LITE_CONFIGS = {
    "ocr": {
        "text_detector": {"from_pretrained": False},
        "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                            "batch_bucketing": True},
    },
    "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}},
}
CKPT = dict(det=dict(seed=1234, out_bias=-3.0), rec=dict(seed=1235, eos_bias=5.0), lay=dict(seed=1240, num_classes=6, score_bias=-2.0),
            tab=dict(seed=1241, num_classes=3, score_bias=-1.0))


def make_checkpoints():
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    return {"det": dbnet_state_dict(**CKPT["det"]), "rec": parseq_state_dict(**CKPT["rec"]),
            "lay": rtdetr_state_dict(**CKPT["lay"]), "tab": rtdetr_state_dict(**CKPT["tab"])}


def build_analyzer(device, sds):
    import logging

    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd.schemas import LayoutAnalyzerSchema, TextDetectorSchema

    logging.getLogger("yomitoku_amd.base").setLevel(logging.WARNING)

    class TruthDrivenAnalyzer(da.DocumentAnalyzer):
        """DocumentAnalyzer whose recogniser / table stages consume ground-truth units (see module doc)."""

        truth_quads = None
        truth_tables = None

        def _detect_and_recognize(self, page):
            self.text_detector(page)  # full detector stage; its noise boxes are not propagated
            det = TextDetectorSchema(points=self.truth_quads, scores=[1.0] * len(self.truth_quads))
            rec, ocr = self.text_recognizer(page, det.points, None)
            return det, rec, ocr

    class TruthLayout:
        def __init__(self, inner, owner):
            self.inner, self.owner = inner, owner

        def __call__(self, page):
            layout_results, _ = self.inner.layout_parser(page)
            tables, _ = self.inner.table_structure_recognizer(page, self.owner.truth_tables)
            return LayoutAnalyzerSchema(paragraphs=layout_results.paragraphs, tables=tables, figures=layout_results.figures), None

    an = TruthDrivenAnalyzer(configs=LITE_CONFIGS, device=str(device))
    an.text_detector.model.load_state_dict(sds["det"])
    an.text_recognizer.model.load_state_dict(sds["rec"])
    an.layout.layout_parser.model.load_state_dict(sds["lay"])
    an.layout.table_structure_recognizer.model.load_state_dict(sds["tab"])
    an.layout = TruthLayout(an.layout, an)
    return an


def cpu_analyzer_page(sds, img, quads, tables, charset):
    """The same page through the oracle chain on the host cores (what `--lite -d cpu` computes)."""
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg

    op.detect(sds["det"], img)
    ocfg = make_cfg(**PRESETS["parseq-tiny-dynw-v4"])
    op.recognize(sds["rec"], ocfg, img, quads, charset, dynamic_width=True, batch_bucketing=True, width_budget=8000,
                 max_batch_size=64, batch_size=10)
    op.layout(sds["lay"], img)
    op.tables(sds["tab"], img, tables)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="analyzer", choices=["analyzer", "detector"])
    ap.add_argument("--pages", type=int, default=8, help="pages per step (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from yomitoku_amd import _lib
    from yomitoku_amd import distributed as ydist
    from yomitoku_amd import imaging
    from yomitoku_amd.utils.synth import synthetic_page_with_truth

    rank, local_rank, world = ydist.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    lib = _lib.load()

    # ---- weights: drawn once on rank 0, ONE flat RCCL broadcast per checkpoint over xGMI
    sds = make_checkpoints() if rank == 0 else {k: None for k in ("det", "rec", "lay", "tab")}
    for k in ("det", "rec", "lay", "tab"):
        sds[k] = ydist.broadcast_state_dict(sds[k], src=0, device=device)

    # ---- synthetic pages of this rank, resident in HBM before the clock starts
    pages = [synthetic_page_with_truth(1000 * rank + i) for i in range(args.pages)]
    pages_dev = [imaging.page_to_device(p[0], device) for p in pages]
    n_lines = [len(p[1]) for p in pages]
    n_tables = [len(p[2]) for p in pages]

    extra = {}
    if args.workload == "analyzer":
        an = build_analyzer(device, sds)

        def step():
            out = None
            for (img, quads, tables), pdev in zip(pages, pages_dev):
                an.truth_quads, an.truth_tables = quads, tables
                out = an(pdev)[0]
            return out

        metric = "pages/sec (DocumentAnalyzer @1600x1200, lite model set)"
        workload = (f"Full DocumentAnalyzer, one page at a time: DBNet (dbnetv2_1) + PARSeq parseq-tiny-dynw-v4 "
                    f"(dynamic_width, batch_bucketing) + RT-DETRv2 layout + RT-DETRv2 table structure + host post-processing "
                    f"and aggregation; {args.pages} synthetic 1600x1200 pages per step per GPU (BASELINE.json configs[3]); "
                    f"recogniser / table stages driven by ground-truth units ({np.mean(n_lines):.0f} text lines, "
                    f"{np.mean(n_tables):.1f} tables per page) because seeded random weights detect noise")
    else:
        from yomitoku_amd.nets import DBNet

        net = DBNet().load_state_dict(sds["det"]).to(device)
        x = torch.cat([imaging.detector_tensor(p, 1280, 1600) for p in pages_dev], 0)

        def step():
            return net(x)["binary"]

        metric = "pages/sec (TextDetector DBNet forward @1600x1200 -> 3x1600x1184)"
        workload = f"TextDetector DBNet forward alone, batch={args.pages} synthetic 1600x1200 pages per GPU (BASELINE.json configs[1])"

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
    assert out is not None

    # ---- roofline leg: per-launch HIP events around the conv kernel (single thread: serial page loop)
    roof = None
    if rank == 0:
        if args.workload == "analyzer":
            an._pool.shutdown(wait=True)
            from concurrent.futures import ThreadPoolExecutor

            an._pool = ThreadPoolExecutor(max_workers=1)  # the event bookkeeping is single-threaded
        _lib.check(lib.ymk_prof_begin())
        step()
        torch.cuda.synchronize()
        ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
        if ms.value > 0:
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            roof = {
                "bound": "mfma",
                "kernel": "conv_igemm (fp32 MFMA implicit GEMM: every conv / linear layer of the four nets)",
                "achieved": round(achieved, 2),
                "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": None,
                "launches_per_step": int(ln.value),
                "avg_launch_us": round(ms.value * 1e3 / max(1, ln.value), 2),
                "kernel_ms_per_step": round(ms.value, 3),
                "gflop_per_step": round(fl.value / 1e9, 1),
            }

    # ---- CPU baseline leg (rank 0, N=1): oracle chain on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        t1 = time.perf_counter()
        n_cpu = 0
        if args.workload == "analyzer":
            charset = an.text_recognizer.charset
            while n_cpu < 1 or (time.perf_counter() - t1 < 12.0 and n_cpu < 3):
                img, quads, tables = pages[n_cpu % len(pages)]
                cpu_analyzer_page(sds, img, quads, tables, charset)
                n_cpu += 1
            sample = (f"{n_cpu} of the same synthetic pages through the oracle restatement (PyTorch-CPU fp32) of the `--lite` "
                      "chain: detector + recogniser + layout + table nets with their pre/post-processing, "
                      "without source_downscale / onnxruntime")
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
            "config": {"workload": workload, "pages_per_step_per_gpu": args.pages, "parallelism": f"page-sharded x{world}",
                       "checkpoints": "seeded synthetic (no network)", **extra},
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

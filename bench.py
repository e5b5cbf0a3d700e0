#!/usr/bin/env python
"""Benchmark of the MI355X DocumentAnalyzer hot path (contract: task prompt / DESIGN.md §8).

    python bench.py --gpus N --steps K --warmup W [--workload analyzer|detector|recognizer] [--model-set lite|default]
                    [--pages 64] [--wave 8] [--in-flight 3]

One rank = ONE process per GPU.  Under torchrun the rank comes from the environment; `python bench.py --gpus N` without
one spawns its own N ranks (same protocol: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  A "step" is one pass of the hot
path over one batch of synthetic 1600x1200 pages (uint8 BGR, as `cv2.imread` would hand them over); the analyzer
workload starts from HOST pages - pinned staging and the H2D copies are inside the clock.  Rank 0 prints ONE JSON line: BASELINE.json's metric (pages/s, whole job), the roofline of the
dominant kernel (live HIP-event timing of every implicit-GEMM convolution launch on its own stream, in a serial pass
of the same program) and, at N=1, the CPU baseline (oracle restatement of the reference's PyTorch-CPU path).

workload analyzer (default; BASELINE.json configs[3]): DBNet text detector, PARSeq recogniser, RT-DETRv2 layout
  parser, RT-DETRv2 table-structure recogniser, host post-processing and aggregation, through the product's multi-page
  entry point `DocumentAnalyzer.serve(host_pages)` (yomitoku_amd/serving.py): `--wave` pages share device batches
  (DBNet / RT-DETR forwards over the wave, one grouped PARSeq forward with one greedy loop), a stage pipeline keeps
  `--in-flight` waves between upload and aggregation, and EVERY page's DocumentAnalyzerSchema comes back, in page
  order, inside the clock.  Per-page results equal `DocumentAnalyzer.__call__` page by page (tests/test_serving_gpu.py,
  tests/test_pipeline_gpu.py).  `--model-set lite` = the reference's `--lite` switches (cli/main.py:505-520:
  parseq-tiny-dynw-v4, dynamic_width, batch_bucketing, source_downscale); `--model-set default` = the constructor
  defaults (dbnetv2_1 + parseq-large-v4_1 at its fixed 800 px canvas).
  Weights are seeded random draws (no network) which detect noise, so the DISCRETE hand-overs between stages use the
  page generator's ground truth - the recogniser gets the true text-line quads, the table recogniser the true table
  boxes, the aggregation the true paragraph boxes - and the DB box extraction runs on a probability map RENDERED from
  the true quads (so that ~80 boxes per page are traced, scored and unclipped inside the clock), while every network
  and every pre/post stage still runs at full cost on every page inside the timed region.
workload detector (configs[1]): DBNet forward alone on a batch of 8 pages.
workload recognizer (configs[2]): TextRecognizer (`--rec-model parseq` = open-beta geometry, or the lite model) on 2048
  synthetic text lines per step; metric text-lines/sec.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")  # as `import yomitoku_amd` does; here too because torch is imported first

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
F16_MFMA_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 / f16 MFMA (v_mfma_f32_32x32x16_f16)
# what the chip holds under back-to-back 16-bit MFMAs and nothing else: 32 cycles per v_mfma_f32_32x32x16_f16 at the 1.74 GHz
# it clocks down to (tools/diag/mfma_valu_overlap.hip, profiles/r05_conv_dma_ablation.md): 256 CUs x 4 x 1024 FLOP/clock.
# Reported NEXT to the guide's peak, never instead of it.
F16_MFMA_SUSTAINED_TFLOPS = 1820.0
HBM_PEAK_GBS, HBM_ACHIEVABLE_GBS = 8000.0, 6290.0  # MI355X_MICROARCH.md: HBM3E spec / measured float4 copy
# "conv_split" code (include/ymk.h) -> (16-bit MFMA products per fp32-grade product, dtype string of the bench line)
SPLIT_MODES = {
    0: (0, "f32 (v_mfma_f32_32x32x2_f32: exact fp32 products, an fmaf chain in k order)"),
    16: (3, "f32 held as two scaled fp16 planes per operand on the MFMA pipe: 3 x v_mfma_f32_32x32x16_f16 per product tile (products to "
            "2^-21, below the fp32 accumulation's own rounding), fp32 accumulate, fp32 activations in HBM; attention, LayerNorm, the "
            "fused greedy step and every grid-starved launch exact fp32"),
    2: (3, "f32 held as two bf16 planes per operand (products to 2^-15): 3 x v_mfma_f32_32x32x16_bf16 per product tile, fp32 accumulate"),
    3: (6, "f32 held as three bf16 planes per operand (products to 2^-23): 6 x v_mfma_f32_32x32x16_bf16 per product tile, fp32 accumulate"),
}


def split_mode():
    """The operand form the models of this process run with: YMK_CONV_SPLIT / YMK_DEBUG_OPTIONS if set, else the library's
    default for models (SPLIT_MODEL_DEFAULT in ymk_common.h: 16)."""
    v = os.environ.get("YMK_CONV_SPLIT")
    for item in filter(None, os.environ.get("YMK_DEBUG_OPTIONS", "").split(",")):
        k, _, val = item.partition("=")
        if k.strip() == "conv_split":
            v = val
    v = 16 if v is None or int(v) < 0 else int(v)
    return v if v in SPLIT_MODES else 0


def kernel_source_sha():
    """Hash of the convolution kernels' sources: stamps PMC-derived numbers kept under profiles/ (stale once a kernel changes)."""
    import hashlib

    h = hashlib.sha256()
    for name in ("ymk_conv.hip", "ymk_conv_split.hip", "ymk_conv_dma.hip", "ymk_conv_astat.hip", "ymk_vit_mlp.hip", "ymk_conv_kernel.h"):
        with open(os.path.join(ROOT, "yomitoku_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]

# YMK_BENCH_DRY=1: CPU rehearsal of the ORCHESTRATION only (ranks over gloo, helper processes, step barriers, the
# max-over-ranks clock, teardown) with stub page workers from tests/bench_dry_stubs.py - what tests/test_bench_dry.py
# runs at world size 2 and 8, since multi-GPU boxes are the driver's.  It does no GPU work and its line says "dry_run".
DRY = os.environ.get("YMK_BENCH_DRY") == "1"


class HighWater:
    """Device memory in use (hipMemGetInfo: torch's allocator AND the library's own arenas), this process's resident set and
    the host's available memory, sampled at every leg boundary and twice a second in between by a side thread, folded per bench
    leg (VERDICT round 4, weak point 8: "no VRAM / RSS high-water measurement of bench.py's legs").  `mark(leg)` names what
    runs from now on.  The interval is coarse on purpose: hipMemGetInfo goes through the driver, and the peaks it is after are
    the models' reserved workspaces, which stand for the whole leg.  YMK_HIGHWATER_INTERVAL=0 turns the side thread off."""

    def __init__(self):
        import threading

        self.legs, self.order, self.leg = {}, [], None
        self.lock, self.stop = threading.Lock(), threading.Event()
        free, total = torch.cuda.mem_get_info()
        self.total, self.before = int(total), int(total - free)
        self.interval = float(os.environ.get("YMK_HIGHWATER_INTERVAL", 0.5))
        self.thread = threading.Thread(target=self._loop, name="ymk-bench-highwater", daemon=True)
        if self.interval > 0:
            self.thread.start()

    @staticmethod
    def _host():
        rss = avail = 0
        try:
            with open("/proc/self/statm") as f:
                rss = int(f.read().split()[1]) * os.sysconf("SC_PAGE_SIZE")
            with open("/proc/meminfo") as f:
                for ln in f:
                    if ln.startswith("MemAvailable:"):
                        avail = int(ln.split()[1]) * 1024
                        break
        except OSError:
            pass
        return rss, avail

    def _sample(self):
        try:
            free, total = torch.cuda.mem_get_info()
        except Exception:  # noqa: BLE001
            return
        rss, avail = self._host()
        with self.lock:
            m = self.legs.get(self.leg)
            if m is not None:
                m["vram"] = max(m["vram"], int(total - free))
                m["rss"] = max(m["rss"], rss)
                m["avail"] = min(m["avail"], avail) if m["avail"] else avail

    def _loop(self):
        while not self.stop.wait(self.interval):
            self._sample()

    def mark(self, leg):
        self._sample()
        with self.lock:
            self.leg = leg
            if leg not in self.legs:
                self.legs[leg] = {"vram": 0, "rss": 0, "avail": 0, "t0": time.perf_counter()}
                self.order.append(leg)
        self._sample()

    def report(self):
        self.mark("end")
        self.stop.set()
        if self.interval > 0:
            self.thread.join(timeout=2)
        gb = float(1 << 30)
        legs = {}
        for a, b in zip(self.order, self.order[1:]):
            m = self.legs[a]
            legs[a] = {"vram_peak_gb": round(m["vram"] / gb, 2), "rss_peak_gb": round(m["rss"] / gb, 2),
                       "host_available_min_gb": round(m["avail"] / gb, 1), "seconds": round(self.legs[b]["t0"] - m["t0"], 1)}
        return {"device_total_gb": round(self.total / gb, 1), "device_used_before_gb": round(self.before / gb, 2),
                "vram_peak_gb": max((v["vram_peak_gb"] for v in legs.values()), default=0.0),
                "rss_peak_gb": max((v["rss_peak_gb"] for v in legs.values()), default=0.0), "legs": legs,
                "note": "hipMemGetInfo (everything on the device) and /proc/self/statm at every leg boundary and every "
                        f"{self.interval:g} s in between; per leg of this process"}


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
DEFAULT_CONFIGS = {  # the constructor defaults of the reference modules (text_detector.py:37, text_recognizer.py:51)
    "ocr": {"text_detector": {"from_pretrained": False}, "text_recognizer": {"from_pretrained": False}},
    "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}},
}
MODEL_SETS = {"lite": LITE_CONFIGS, "default": DEFAULT_CONFIGS}
# Biases of the seeded heads, chosen so that the random nets emit realistic unit COUNTS: sparse blobs
# above the DB threshold, text of ~10-30 tokens per line batch, a handful of layout / row / column boxes.
CKPT = dict(det=dict(seed=1234), rec=dict(seed=1235, eos_bias=6.1), lay=dict(seed=1240, num_classes=6, score_gain=3.0),
            tab=dict(seed=1241, num_classes=3, score_gain=3.0))
REC_CKPT_OF_SET = {  # synth.parseq_state_dict kwargs per model set, oracle preset, oracle batching kwargs
    "lite": (dict(seed=1235, eos_bias=6.1), "parseq-tiny-dynw-v4",
             dict(dynamic_width=True, batch_bucketing=True, width_budget=8000, max_batch_size=64, batch_size=10, source_downscale=True)),
    "default": (dict(seed=1237, patch=(8, 8), enc_dim=768, dec_dim=768, num_tokens=7121, eos_bias=6.5), "parseq-large-v4_1",
                dict(dynamic_width=False, batch_bucketing=False, width_budget=None, max_batch_size=None, batch_size=128)),
}

if os.environ.get("YMK_BENCH_CKPT"):  # e.g. '{"rec": {"eos_bias": 5.5}}' - for unit-count sweeps
    for _k, _v in json.loads(os.environ["YMK_BENCH_CKPT"]).items():
        CKPT[_k].update(_v)


def make_checkpoints(model_set="lite"):
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    rec_kw = dict(CKPT["rec"]) if model_set == "lite" else dict(REC_CKPT_OF_SET[model_set][0])
    return {"det": dbnet_state_dict(**CKPT["det"]), "rec": parseq_state_dict(**rec_kw),
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


def render_truth_map(quads, page_hw, map_hw):
    """Probability map a trained DBNet would emit for the page's true text lines: each quad shrunk by ~1/6 of its height
    (the DB training target), filled with 0.93, edges softened by a 5 x 5 box blur.  Pre-rendered per page before the
    clock; inside the clock it is what ymk_db_postprocess traces, scores and unclips."""
    from scipy.ndimage import uniform_filter

    (ph, pw), (mh, mw) = page_hw, map_hw
    m = np.zeros((mh, mw), dtype=np.float32)
    sy, sx = mh / ph, mw / pw
    for q in quads:
        xs, ys = [p[0] for p in q], [p[1] for p in q]
        x0, x1, y0, y1 = min(xs) * sx, max(xs) * sx, min(ys) * sy, max(ys) * sy
        d = max(1.0, (y1 - y0) / 6.0)
        a, b, c, e = int(round(y0 + d)), int(round(y1 - d)), int(round(x0 + d)), int(round(x1 - d))
        if b > a and e > c:
            m[a:b, c:e] = 0.93
    return np.ascontiguousarray(uniform_filter(m, size=5, mode="constant"))


class Page:
    def __init__(self, seed, device):
        from yomitoku_amd import imaging
        from yomitoku_amd.utils.synth import synthetic_page_with_truth

        self.img, self.quads, self.tables, self.paragraphs = synthetic_page_with_truth(seed)
        self.dev = imaging.page_to_device(self.img, device)
        h, w = self.img.shape[:2]
        self.truth_map = render_truth_map(self.quads, (h, w), imaging.resize_shortest_edge_dims(h, w, 1280, 1600))


class _TruthView:
    """`an.truth = pages` / `an.stats = {...}` of the tools and legs of this file, forwarded to the analyzer's handover."""

    def __init__(self, key):
        self.key = key

    def __get__(self, obj, owner=None):
        return None if obj is None else getattr(obj.handover, self.key)

    def __set__(self, obj, value):
        setattr(obj.handover, self.key, value)


def build_analyzer(device, sds, model_set="lite"):
    """The analyzer of this rank: the product's DocumentAnalyzer - its own stage bodies, nothing overridden - whose DISCRETE
    stage hand-overs carry the pages' ground truth through the product's documented hook (DocumentAnalyzer.handover,
    yomitoku_amd/testing.py: TruthHandover).  `an.truth` is the list of Page objects the served pages come from (page id i ->
    truth[i % len(truth)])."""
    import logging

    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd.testing import TruthHandover

    logging.getLogger("yomitoku_amd.base").setLevel(logging.WARNING)

    class TruthDrivenAnalyzer(da.DocumentAnalyzer):  # no method of the product is overridden: two attribute views, nothing else
        truth = _TruthView("truth")
        stats = _TruthView("stats")

    an = TruthDrivenAnalyzer(configs=MODEL_SETS[model_set], device=str(device))
    an.handover = TruthHandover()
    an.text_detector.model.load_state_dict(sds["det"])
    an.text_recognizer.model.load_state_dict(sds["rec"])
    an.layout.layout_parser.model.load_state_dict(sds["lay"])
    an.layout.table_structure_recognizer.model.load_state_dict(sds["tab"])
    return an


def cpu_analyzer_page(sds, page: Page, charset, model_set="lite"):
    """The same page through the oracle chain on the host cores (what `-d cpu` computes with this model set)."""
    from oracle import pipeline as op
    from oracle.parseq import PRESETS, make_cfg

    _, preset, batching = REC_CKPT_OF_SET[model_set]
    op.detect(sds["det"], page.img)
    ocfg = make_cfg(**PRESETS[preset])
    op.recognize(sds["rec"], ocfg, page.img, page.quads, charset, **batching)
    op.layout(sds["lay"], page.img)
    op.tables(sds["tab"], page.img, page.tables)


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


def conv_roofline(lib, run_once, units, unit_name, kernel_desc, reps=3, split=None):
    """HIP events around every implicit-GEMM launch of one serial pass (`run_once`) -> the roofline dict.  The pass is
    repeated `reps` times and the MEDIAN pass (by total event time) is reported, with the spread next to it."""
    from yomitoku_amd import _lib

    passes = []
    for _ in range(max(1, reps)):
        _lib.check(lib.ymk_prof_begin())
        run_once()
        torch.cuda.synchronize()
        ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
        alg = ctypes.c_double()
        _lib.check(lib.ymk_prof_bytes(ctypes.byref(alg)))
        if ms.value > 0:
            passes.append((ms.value, fl.value, ln.value, alg.value, _lib.prof_launch_table()))
    if not passes:
        return None
    passes.sort(key=lambda p: p[0])
    ms, fl, ln, alg, table = passes[len(passes) // 2]
    ach = fl / (ms * 1e-3) / 1e12
    products = SPLIT_MODES[split if split is not None else split_mode()][0]
    # the two roofs of the implicit-GEMM kernels.  MFMA: exact fp32 MFMA, or the 16-bit MFMA rate over the MFMA products one
    # fp32-grade product costs (algorithmic FLOPs = 2 M N K throughout: fp32-equivalent work per second).  HBM: the
    # algorithmic bytes of a launch (input view, weights, output and residual, each touched once - fp32 activations in either
    # form) over its duration, against the guide's 8 TB/s (6.29 TB/s measured achievable).  The line's top-level
    # bound / achieved / peak / frac are those of the roof the path sits CLOSER to (the binding one); both views are kept.
    mfma_peak = FP32_MFMA_PEAK_TFLOPS if products == 0 else F16_MFMA_PEAK_TFLOPS / products
    gbs = alg / (ms * 1e-3) / 1e9
    mfma = {"achieved": round(ach, 2), "peak": round(mfma_peak, 1), "unit": "TFLOP/s", "frac": round(ach / mfma_peak, 4),
            "peak_note": ("dense fp32 MFMA (v_mfma_f32_32x32x2_f32)" if products == 0 else
                          f"dense 16-bit MFMA {F16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s / {products} MFMA products per fp32-grade product = fp32-equivalent "
                          f"TFLOP/s; the same `achieved` is {ach / FP32_MFMA_PEAK_TFLOPS:.3f} of the exact-fp32 MFMA roof (157.3) of the round-3 line")}
    if products:
        mfma["frac_of_sustained"] = round(ach / (F16_MFMA_SUSTAINED_TFLOPS / products), 4)
        mfma["sustained_note"] = (f"{F16_MFMA_SUSTAINED_TFLOPS:.0f} TFLOP/s: the 16-bit MFMA rate measured with nothing but MFMAs in flight (the clock drops to "
                                  "1.74 GHz under them; profiles/r05_conv_dma_ablation.md) - context for `frac`, which stays against the guide's peak")
    hbm = {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
           "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4),
           "peak_note": f"HBM3E {HBM_PEAK_GBS:.0f} GB/s spec; {HBM_ACHIEVABLE_GBS:.0f} GB/s measured achievable (MI355X_MICROARCH.md); algorithmic bytes, not PMC traffic"}
    top = dict(hbm, bound="hbm") if hbm["frac"] > mfma["frac"] else dict(mfma, bound="mfma")
    return {
        "bound": top["bound"], "kernel": kernel_desc, "achieved": top["achieved"], "peak": top["peak"], "unit": top["unit"],
        "frac": top["frac"], "traffic": None, "mfma": mfma, "hbm": hbm, "achieved_tflops": round(ach, 2),
        "frac_of_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
        "algorithmic_bytes_per_launch": int(alg // max(1, ln)),
        f"launches_per_{unit_name}": round(ln / units, 2), "avg_launch_us": round(ms * 1e3 / max(1, ln), 2),
        f"kernel_ms_per_{unit_name}": round(ms / units, 4), f"gflop_per_{unit_name}": round(fl / units / 1e9, 2),
        "serial_passes_tflops": [round(p[1] / (p[0] * 1e-3) / 1e12, 2) for p in passes],
        "per_launch": two_roof_bound(table),
    }


def two_roof_bound(table):
    """Every launch of the pass priced against ITS binding roof: t_bound = max(FLOPs / MFMA peak of the kernel it ran on,
    algorithmic bytes / HBM peak); sum of the bounds over sum of the measured launch times = how far the path as a whole
    is from the two roofs taken together (the single-roof fractions above divide totals, which undersells a mix of
    HBM-bound 1 x 1 layers and MFMA-bound k x k layers).  `table`: _lib.prof_launch_table() rows."""
    if not table:
        return None
    t_meas = t_bound = t_bound_ach = t_hbm_side = 0.0
    n_hbm = 0
    for ms, flop, nbytes, products in table:
        peak = FP32_MFMA_PEAK_TFLOPS if products == 0 else F16_MFMA_PEAK_TFLOPS / products
        t_m = flop / (peak * 1e12) * 1e3
        t_h = nbytes / (HBM_PEAK_GBS * 1e9) * 1e3
        t_meas += ms
        t_bound += max(t_m, t_h)
        t_bound_ach += max(t_m, nbytes / (HBM_ACHIEVABLE_GBS * 1e9) * 1e3)
        if t_h > t_m:
            n_hbm += 1
            t_hbm_side += ms
    return {"launches": len(table), "measured_ms": round(t_meas, 3), "two_roof_bound_ms": round(t_bound, 3),
            "frac_of_two_roof_bound": round(t_bound / t_meas, 4),
            "frac_with_achievable_hbm": round(t_bound_ach / t_meas, 4),
            "hbm_bound_launches": n_hbm, "hbm_bound_share_of_measured_time": round(t_hbm_side / t_meas, 4),
            "note": "per launch: max(algorithmic FLOPs / MFMA peak of the kernel that ran it, algorithmic bytes / 8 TB/s); "
                    "frac = sum of those bounds / sum of the measured launch times"}


def cpu_timed(fn, n_warm, n_timed, budget_s):
    """n_warm untimed + up to n_timed timed repetitions of fn(i) (stops early past budget_s, never below 1 timed)."""
    for i in range(n_warm):
        fn(i)
    times = []
    t_all = time.perf_counter()
    for i in range(n_timed):
        t = time.perf_counter()
        fn(n_warm + i)
        times.append(time.perf_counter() - t)
        if time.perf_counter() - t_all > budget_s:
            break
    return times


def recognizer_setup(device, rec_model, lines, seed=1, sd=None):
    """TextRecognizer (`rec_model` geometry, seeded weights) + one synthetic sheet of `lines` text lines resident in HBM."""
    from yomitoku_amd import imaging
    from yomitoku_amd.text_recognizer import TextRecognizer
    from yomitoku_amd.utils.synth import parseq_state_dict, synthetic_line_sheet

    ckpt_kw, _, _ = REC_PRESETS[rec_model]
    if sd is None:
        sd = parseq_state_dict(**ckpt_kw)
    sheet, quads = synthetic_line_sheet(seed=seed, n_lines=lines)
    page = imaging.page_to_device(sheet, device)
    rec = TextRecognizer(model_name=rec_model, from_pretrained=False, device=str(device), dynamic_width=True, batch_bucketing=True)
    rec.model.load_state_dict(sd)
    return rec, sd, sheet, page, quads


def recognizer_workload(args, rank, local_rank, world, device, lib):
    """BASELINE.json configs[2]: TextRecognizer with dynamic_width + batch_bucketing on 2048 synthetic 32 x W text-line
    crops per GPU per step (one sheet image + 2048 quads through TextRecognizer.__call__).  Unit: text lines."""
    from yomitoku_amd import distributed as ydist
    from yomitoku_amd.utils.synth import parseq_state_dict

    ckpt_kw, preset, batching = REC_PRESETS[args.rec_model]
    sd = ydist.broadcast_state_dict(parseq_state_dict(**ckpt_kw) if rank == 0 else None, src=0, device=device)
    rccl = ydist.replica_report({"rec": sd}, device)
    rec, sd, sheet, page, quads = recognizer_setup(device, args.rec_model, args.lines, seed=1 + rank, sd=sd)

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
    cpu = None
    roof = conv_roofline(lib, step, args.lines, "line", "conv_igemm / conv_splitk (every linear layer of the ViT encoder and the decoder)")
    if world == 1 and not args.no_cpu_baseline:
        from oracle import pipeline as op
        from oracle.parseq import PRESETS, make_cfg

        n_cpu = min(args.lines, 96)
        ocfg = make_cfg(**PRESETS[preset])
        times = cpu_timed(lambda i: op.recognize(sd, ocfg, sheet, quads[:n_cpu], rec.charset, dynamic_width=True, batch_bucketing=True,
                                                 **batching), 1, 3, 40.0)
        cpu = {"value": round(n_cpu / float(np.median(times)), 2), "unit": "lines/s", "cores": torch.get_num_threads(), "kind": "port",
               "runs": [round(n_cpu / t, 2) for t in times],
               "sample": f"the first {n_cpu} of the same lines through oracle.pipeline.recognize (PyTorch-CPU fp32 restatement), "
                         f"1 warm-up + {len(times)} timed passes, median"}
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
                   "parallelism": f"line sheets sharded x{world} GPU(s); the mini-batches of a call share grouped forwards of "
                                  f"<= {rec.MAX_LINES_PER_FORWARD} lines",
                   "checkpoints": "seeded synthetic (no network)", "last_forward_ar_steps": int(rec.model.last_ar_steps)},
        "roofline": roof, "cpu_baseline": cpu, "rccl": rccl,
    }


def secondary_metrics(args, device, sds, pages):
    """The other numbers BASELINE.json's metric names, measured in this process AFTER the timed region (rank 0, N = 1) so
    that the driver's one bench line carries them: PARSeq text-lines/sec (configs[2]) for the open-beta geometry and for
    the --lite recogniser, and the analyzer with the reference's DEFAULT model set.  Short legs: 2 warm-up + 3 timed
    steps of 2048 lines; two warm-up waves + two timed passes over the pages as one job (timed_serve)."""
    out = {}
    for rec_model in ("parseq", "parseq-tiny-dynw-v4"):
        rec, _, _, page, quads = recognizer_setup(device, rec_model, 2048)
        for _ in range(2):  # shapes, workspace growth
            rec(page, quads)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            res = rec(page, quads)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[f"lines_per_s_{rec_model}"] = {"value": round(3 * 2048 / dt, 1), "unit": "lines/s", "steps": 3, "lines_per_step": 2048,
                                           "workload": "BASELINE.json configs[2]: one TextRecognizer call (dynamic_width, batch_bucketing) "
                                                       "over 2048 synthetic 32 x W lines of one sheet resident in HBM",
                                           "distinct_strings": len(set(res.contents))}
        rec.model.close()
        del rec
    sds_def = dict(sds, rec=make_checkpoints("default")["rec"])
    an = build_analyzer(device, sds_def, "default")
    an.truth = pages
    host = [p.img for p in pages]
    # waves of 8 pages here: parseq-large-v4_1 pads every line to 800 px (400 tokens of 768), so a wave of 8 pages already is a
    # 650-line forward of 260 000 tokens; waves of 16 measured 18.5 against 23.8 pages/s
    import copy

    args8 = copy.copy(args)
    args8.wave = min(8, args.wave)
    res, dt = timed_serve(args8, an, host)
    out["pages_per_s_default_model_set"] = {"value": round(len(res) / dt, 2), "unit": "pages/s", "steps": len(res) // len(host), "pages": len(res), "wave": args8.wave,
                                            "failed_pages": sum(isinstance(r, BaseException) for r in res),
                                            "workload": "the analyzer workload with the reference's constructor defaults: dbnetv2_1 + "
                                                        "parseq-large-v4_1 (fixed 800 px canvas, batch 128) + RT-DETRv2 layout + table"}
    close_analyzer(an)
    return out


def timed_serve(args, an, host_pages, warm=None, steps=2):
    """Secondary legs: one warm-up job of two waves, then `steps` passes over the pages as ONE timed job (as the headline's
    steps are: a single 64-page job of four 16-page waves would mostly measure the pipeline filling and draining)."""
    warm = 2 * args.wave if warm is None else warm
    an.serve(host_pages[:warm], wave=args.wave, in_flight=args.in_flight)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = an.serve(host_pages * steps, wave=args.wave, in_flight=args.in_flight)
    torch.cuda.synchronize()
    return res, time.perf_counter() - t0


def analyzer_nets(an):
    return (an.text_detector.model, an.text_recognizer.model, an.layout.layout_parser.model, an.layout.table_structure_recognizer.model)


def close_analyzer(an):
    """The pipeline threads, the recogniser's replica handles AND the four nets' device memory (weights + reserved workspaces,
    tens of GB per analyzer) - now, not when the garbage collector gets to the object (`serve` defers full collections)."""
    an.close()
    for n in analyzer_nets(an):
        n.close()


def exact_fp32_metrics(args, an, host_pages):
    """The SAME analyzer and pages with every net on the exact fp32 MFMA kernels ("conv_split" 0) - the round-3 headline
    configuration, kept as the yardstick next to the fp16-split default.  Two warm-up waves, FIVE timed passes over the pages as one job."""
    for n in analyzer_nets(an):
        n.set_conv_split(0)
    try:
        res, dt = timed_serve(args, an, host_pages, steps=5)
    finally:
        for n in analyzer_nets(an):
            n.set_conv_split(None)
    return {"value": round(len(res) / dt, 2), "unit": "pages/s", "steps": len(res) // len(host_pages), "pages": len(res), "dtype": SPLIT_MODES[0][1],
            "failed_pages": sum(isinstance(r, BaseException) for r in res)}


def unmodified_serve_metrics(args, device, sds, host_pages):
    """`DocumentAnalyzer.serve` exactly as the product ships it - no ground-truth hand-overs: the calibrated seeded heads'
    own detections flow from stage to stage, so the product's own _stage_boxes / _stage_tables / _stage_cells bodies are inside
    the clock.  What the nets detect is noise (seeded weights), so the unit counts differ from the headline's; they are
    reported next to the rate.  Two warm-up waves, two timed passes as one job."""
    from yomitoku_amd import DocumentAnalyzer
    from yomitoku_amd.utils.synth import dbnet_state_dict

    # the headline's detector checkpoint is calibrated for a map that is 2 % "text": speckle that the box filters reject, so
    # no word would reach the recogniser.  This leg takes the seeded detector whose noise field yields boxes that survive
    # (seed 8, bias -1.5: ~90 on the reference's 596 x 842 test page, tests/test_baseline_configs_gpu.py)
    sds = dict(sds, det=dbnet_state_dict(8, out_bias=-1.5))
    an = DocumentAnalyzer(configs=MODEL_SETS[args.model_set], device=str(device))
    try:
        for net, key in zip(analyzer_nets(an), ("det", "rec", "lay", "tab")):
            net.load_state_dict(sds[key])
        res, dt = timed_serve(args, an, host_pages)
    finally:
        close_analyzer(an)
    ok = [r for r in res if not isinstance(r, BaseException)]
    return {"value": round(len(res) / dt, 2), "unit": "pages/s", "steps": len(res) // len(host_pages), "pages": len(res), "failed_pages": len(res) - len(ok),
            "units_per_page": {"words": round(float(np.mean([len(r.words) for r in ok])), 1) if ok else None,
                               "paragraphs": round(float(np.mean([len(r.paragraphs) for r in ok])), 1) if ok else None,
                               "tables": round(float(np.mean([len(r.tables) for r in ok])), 2) if ok else None,
                               "cells": round(float(np.mean([sum(len(t.cells) for t in r.tables) for r in ok])), 1) if ok else None},
            "workload": "the unmodified product path: DocumentAnalyzer.serve(host pages) with the calibrated seeded checkpoints, every stage "
                        "fed by the previous stage's own output"}


def stage_control_metrics(args, device, sds, pages):
    """Control for the shaped headline workload (VERDICT round 4, weak point 5): the UNMODIFIED stage bodies of the product -
    `_stage_boxes`, `_stage_crops`, `_stage_tables`, `_stage_cells`, `_stage_finish` exactly as `DocumentAnalyzer` ships them -
    at the headline's unit counts.  Seeded weights cannot detect the pages' content, so what is substituted here is the two
    NETWORK OUTPUTS a trained checkpoint would produce, after the networks have run at full cost: the detector's probability
    map (the map rendered from the true lines, the headline's own) and the layout net's raw (logits, boxes) tensors (one
    query per true paragraph / table, score 0.98).  Everything downstream is the product's own code on its own hand-overs:
    DB box extraction -> its boxes to the crop planner and the recogniser, RTDETRPostProcessor + containment filters -> its
    table boxes to the table-structure net, whose (noise, calibrated) rows / columns go through the product's filters and
    cell grids, then aggregation.  Two warm-up waves, two timed passes over the pages as one job."""
    from yomitoku_amd import DocumentAnalyzer

    from yomitoku_amd.testing import NetOutputHandover

    class NetOutputDrivenAnalyzer(DocumentAnalyzer):  # the product's stage bodies; `truth` is a view of the handover's
        truth = _TruthView("truth")

    an = NetOutputDrivenAnalyzer(configs=MODEL_SETS[args.model_set], device=str(device))
    an.handover = NetOutputHandover(categories={c: i for i, c in an.layout.layout_parser.label_mapper.items()})
    try:
        for net, key in zip(analyzer_nets(an), ("det", "rec", "lay", "tab")):
            net.load_state_dict(sds[key])
        an.truth = pages
        res, dt = timed_serve(args, an, [p.img for p in pages])
    finally:
        close_analyzer(an)
    ok = [r for r in res if not isinstance(r, BaseException)]
    return {"value": round(len(res) / dt, 2), "unit": "pages/s", "steps": len(res) // len(pages), "pages": len(res), "failed_pages": len(res) - len(ok),
            "units_per_page": {"words": round(float(np.mean([len(r.words) for r in ok])), 1) if ok else None,
                               "paragraphs": round(float(np.mean([len(r.paragraphs) for r in ok])), 1) if ok else None,
                               "tables": round(float(np.mean([len(r.tables) for r in ok])), 2) if ok else None,
                               "cells": round(float(np.mean([sum(len(t.cells) for t in r.tables) for r in ok])), 1) if ok else None},
            "workload": "control for the headline: the product's own stage bodies (box extraction -> crops, layout post-processing -> table "
                        "crops, table filters and cell grids, aggregation) on their own hand-overs, with the detector map and the layout "
                        "net's raw output replaced - after the forwards - by what a trained net would emit for the page's true lines, "
                        "paragraphs and tables"}


def self_spawn(argv, n):
    """`python bench.py --gpus N` outside torchrun: start the N ranks ourselves (same env protocol), pass rank 0's line
    through, fail if any rank fails."""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), YMK_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    if rc:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="analyzer", choices=["analyzer", "detector", "recognizer"])
    ap.add_argument("--model-set", default="lite", choices=sorted(MODEL_SETS), help="analyzer workload: lite (cli --lite) or default models")
    ap.add_argument("--rec-model", default="parseq", choices=sorted(REC_PRESETS), help="recognizer workload: model (configs[2]: parseq)")
    ap.add_argument("--lines", type=int, default=2048, help="recognizer workload: text lines per step per GPU")
    ap.add_argument("--pages", type=int, default=64, help="pages per step per GPU (BASELINE.json configs[3]: 64)")
    ap.add_argument("--total-pages", type=int, default=0, help="strong scaling (configs[4]: 512): pages per step over ALL GPUs")
    ap.add_argument("--wave", type=int, default=16, help="pages per device batch (analyzer workload; DocumentAnalyzer.serve's default)")
    ap.add_argument("--in-flight", type=int, default=4, help="waves between upload and aggregation (analyzer workload)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary metrics leg (recogniser lines/s, default model set)")
    ap.add_argument("--no-roofline", action="store_true", help="skip the serial roofline pass (pipeline sweeps: only the timed region)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the timed region: only the serial roofline pass (the command profiles/ runs under rocprofv3)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_spawn(sys.argv[1:], args.gpus))

    from yomitoku_amd import _lib
    from yomitoku_amd import distributed as ydist
    from yomitoku_amd import imaging

    rank, local_rank, world = ydist.init("gloo" if DRY else None)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert DRY or torch.cuda.is_available(), "bench.py needs a HIP device"
    device = rank_device(local_rank)
    lib = None if DRY else _lib.load()
    hw = HighWater() if (rank == 0 and not DRY) else None

    def leg(name):
        if hw is not None:
            hw.mark(name)

    leg("setup: checkpoints, broadcast, analyzer, pages")
    if lib is not None and os.environ.get("YMK_DEC_ROWS"):  # A/B knob of the fused greedy step (rows per block)
        _lib.debug_option("dec_rows", int(os.environ["YMK_DEC_ROWS"]))

    if args.workload == "recognizer":
        line = recognizer_workload(args, rank, local_rank, world, device, lib)
        if rank == 0:
            print(json.dumps(line), flush=True)
        if world > 1:
            torch.distributed.barrier()  # rank 0 is still in its roofline / CPU legs: the others wait here, not in teardown
            torch.distributed.destroy_process_group()
        return

    # ---- the sharded job (yomitoku_amd.distributed.ShardedServer): process group, this rank's core slice and thread budget,
    # weights drawn once on rank 0 and broadcast as flat RCCL messages over xGMI, every rank's CRC of what it received, the
    # rank's analyzer.  The timed region below is DocumentAnalyzer.serve on this rank's share, bracketed as the contract asks.
    def rank0_checkpoints():
        sds = make_checkpoints(args.model_set)
        return calibrate_heads(sds, device, Page(0, device)) if args.workload == "analyzer" else sds

    def make_analyzer(dev, sds, budget):
        if args.workload != "analyzer":
            return None
        an = build_analyzer(dev, sds, args.model_set)
        if getattr(an, "text_detector", None) is not None:
            an.text_detector.post_threads = budget["box_threads"]
        return an

    server = ydist.ShardedServer(make_analyzer, rank0_checkpoints, backend="gloo" if DRY else None, device=device, pin_cores=not DRY)
    sds, rccl, an = server.checkpoints, server.replicas, server.analyzer

    # ---- synthetic pages of this rank: HOST arrays for the analyzer (staging + H2D inside the clock)
    if args.workload == "detector":
        args.pages = min(args.pages, 8)
    scaling = "weak"
    if args.total_pages:  # strong scaling: the job's pages dealt round-robin to the ranks (yomitoku_amd.distributed)
        seeds = [1000 + i for i in ydist.shard_indices(args.total_pages, rank, world)]
        scaling = "strong"
    else:
        seeds = [1000 * rank + i for i in range(args.pages)]
    pages_job = args.total_pages if args.total_pages else args.pages * world
    pages = make_pages(seeds, device)
    extra = {"host_threads": dict(server.budget, cores=server.cores)}
    failed = 0
    if args.workload == "analyzer":
        an.truth = pages
        host_pages = [p.img for p in pages]

        def run_steps(k):
            """k steps (k passes over the rank's pages) as ONE serve() job: host pages in, every page's result out, in
            order; the pipeline keeps `in_flight` waves going across the step boundaries (no drain between steps)."""
            nonlocal failed
            res = an.serve(host_pages * int(k), wave=args.wave, in_flight=args.in_flight)
            assert len(res) == len(host_pages) * int(k)
            failed += sum(isinstance(r, BaseException) for r in res)
            return res[-1]

        names = {"lite": "DBNet dbnetv2_1 + PARSeq parseq-tiny-dynw-v4 (dynamic_width, batch_bucketing, source_downscale)",
                 "default": "DBNet dbnetv2_1 + PARSeq parseq-large-v4_1 (fixed 800 px canvas, batch_size 128 per page)"}
        metric = f"pages/sec (DocumentAnalyzer @1600x1200, {args.model_set} model set)"
        workload = (f"Full DocumentAnalyzer (BASELINE.json configs[{4 if args.total_pages else 3}]): {names[args.model_set]} + RT-DETRv2 layout + "
                    f"RT-DETRv2 table structure + host post-processing and aggregation, through DocumentAnalyzer.serve - ONE process per "
                    f"GPU, HOST pages in (pinned staging + H2D inside the clock), every page's DocumentAnalyzerSchema returned in page "
                    f"order; {len(seeds)} synthetic 1600x1200 pages per step on this GPU, in waves of {args.wave} pages (device batches "
                    f"across the pages of a wave), stage pipeline with {args.in_flight} waves in flight, the K steps of the timed region "
                    f"served as one job (waves of step k+1 start while the last waves of step k finish); stage hand-overs use ground "
                    f"truth ({np.mean([len(p.quads) for p in pages]):.0f} text lines, {np.mean([len(p.tables) for p in pages]):.1f} tables, "
                    f"{np.mean([len(p.paragraphs) for p in pages]):.0f} paragraphs per page) and the DB box extraction runs on a map "
                    f"rendered from the true lines, because seeded random weights detect noise")
    else:
        from yomitoku_amd.nets import DBNet

        net = DBNet().load_state_dict(sds["det"]).to(device)
        x = torch.cat([imaging.detector_tensor(p.dev, 1280, 1600) for p in pages], 0)

        def step():
            return net(x)["binary"]

        def run_steps(k):
            out = None
            for _ in range(k):
                out = step()
            return out

        metric = "pages/sec (TextDetector DBNet forward @1600x1200 -> 3x1600x1184)"
        workload = f"TextDetector DBNet forward alone, batch={args.pages} synthetic 1600x1200 pages per GPU (BASELINE.json configs[1])"

    dt = None
    out = None
    per_rank = None
    if not args.roofline_only:
        leg("warm-up steps")
        if args.warmup:
            run_steps(args.warmup)
        failed = 0
        leg("timed region")
        device_sync()
        if world > 1:
            torch.distributed.barrier()
        device_sync()
        t0 = time.perf_counter()
        out = run_steps(args.steps)  # exactly K steps; the analyzer's follow each other without a drain (one serve() job)
        device_sync()
        dt_local = time.perf_counter() - t0
        if world > 1:
            torch.distributed.barrier()
        device_sync()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            dt = float(tt.item())
        assert out is not None
        # every rank's own pages/s (its K steps over its own clock, before the closing barrier) and failed-page count
        rows = ydist.all_gather_scalars([len(seeds) * args.steps / dt_local, failed], device)
        per_rank = {"pages_per_s_min": round(min(r[0] for r in rows), 3), "pages_per_s_max": round(max(r[0] for r in rows), 3),
                    "failed_pages": int(sum(r[1] for r in rows))}

    # ---- roofline leg: per-launch HIP events around the conv kernel, in a SERIAL pass of the same program: the same analyzer,
    # one wave at a time, the two chains of a wave one after the other on one stream - so that a launch's event pair
    # brackets that kernel alone.  `python bench.py --roofline-only` under rocprofv3 is the same pass (profiles/).
    roof = None
    leg("roofline: serial pass with per-launch events")
    if rank == 0 and not DRY and not args.no_roofline:
        kern = ("conv_igemm_split / conv_f16_dma / conv_f16_astat / k_vit_mlp_f16 (fp16-split MFMA implicit GEMM: the chip-filling conv / linear launches of the four nets, the ViT MLP halves as one launch each) + conv_igemm / "
                "conv_splitk (exact fp32 MFMA: the stems and the grid-starved launches), max|x| passes included" if split_mode() else
                "conv_igemm / conv_splitk (fp32 MFMA implicit GEMM: every conv / linear layer of the four nets)")
        if args.workload == "analyzer":
            an.concurrent_chains = False  # the two chains one after the other: an event pair brackets one kernel
            an.stats = {"det_boxes": [], "layout_boxes": [], "cells": []}
            n_prof = min(len(pages), max(args.wave, 16 // args.wave * args.wave))
            prof_pages = [p.dev for p in pages[:n_prof]]  # resident pages: page ids 0.. match an.truth
            units = len(prof_pages)
            an.analyze_pages(prof_pages, wave=args.wave)  # shapes of the serial pass seen once (workspace growth stays out)
            torch.cuda.synchronize()
            an.stats = {"det_boxes": [], "layout_boxes": [], "cells": []}

            def prof_step():
                an.analyze_pages(prof_pages, wave=args.wave)
        else:
            prof_step, units = step, args.pages
        roof = conv_roofline(lib, prof_step, units, "page", kern)
        pmc = next((p for p in (os.path.join(ROOT, "profiles", f"r0{r}_{args.workload}_pmc_conv_traffic.json") for r in (6, 5)) if os.path.exists(p)),
                   os.path.join(ROOT, "profiles", f"r06_{args.workload}_pmc_conv_traffic.json"))  # the newest round's measurement
        if roof is not None and os.path.exists(pmc):
            # HBM bytes per conv launch from the PMC passes of this same serial pass (rocprofv3 cannot run inside bench.py:
            # profiles/README.md has the commands); compare with algorithmic_bytes_per_launch.  The file carries the hash of the
            # kernel sources and the operand form it was measured with: anything else is stale and reported as null.
            with open(pmc) as f:
                t = json.load(f)
            fresh = t.get("kernel_source_sha16") == kernel_source_sha() and int(t.get("conv_split", -1)) == split_mode()
            roof["traffic"] = t["hbm_bytes_per_launch"] if fresh else None
            roof["traffic_source"] = os.path.relpath(pmc, ROOT) + ("" if fresh else " (STALE: measured on other kernel sources / operand form)")
        if roof is not None and dt is not None:
            # the same FLOPs over the WALL clock of the timed region (all kernels, host gaps and overlap included)
            wall_tf = roof["gflop_per_page"] * 1e-3 * pages_job * args.steps / dt / max(1, world)
            roof["wall_implied"] = {"achieved_tflops": round(wall_tf, 2), "frac_of_fp32_mfma_peak": round(wall_tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                    "note": "gflop_per_page x pages/s per GPU: lower than `achieved` because the wall clock also holds "
                                            "the non-conv kernels, copies and host gaps; kernel_ms_per_page x pages_per_step <= ms_per_step "
                                            "is checked below (`self_consistent`)"}
            roof["conv_share_of_wall"] = round(roof["kernel_ms_per_page"] * len(seeds) * args.steps / (dt * 1e3), 4)
            # serial conv time <= wall clock of the timed region, up to what concurrency buys (other waves' kernels fill the
            # last, partly empty block generation of a launch; the serial pass leaves those CUs idle).  Reported, not
            # asserted: a disturbed serial pass must not cost the run its JSON line.
            roof["self_consistent"] = bool(roof["conv_share_of_wall"] <= 1.05)
        if args.workload == "analyzer" and roof is not None:
            # the north star quotes MFMA utilisation "on DBNet conv": the same measurement over the detector's launches alone
            det = an.text_detector
            d_roof = conv_roofline(lib, lambda: det.forward_pages(prof_pages[: args.wave]), len(prof_pages[: args.wave]), "page", kern)
            roof["dbnet_conv"] = {k: d_roof[k] for k in ("bound", "achieved", "peak", "unit", "frac", "achieved_tflops", "mfma", "hbm", "launches_per_page",
                                                         "kernel_ms_per_page", "gflop_per_page")}
            roof["dbnet_conv"]["batch"] = len(prof_pages[: args.wave])
            st = an.stats
            extra["measured_units_per_page"] = {
                "ar_steps_last_forward": int(an.text_recognizer.model.last_ar_steps),
                "db_boxes_extracted": float(np.mean(st["det_boxes"])),
                "noise_layout_boxes": float(np.mean(st["layout_boxes"])),
                "table_cells": float(np.mean(st["cells"])),
            }
            an.concurrent_chains = True
            an.stats = None

    # ---- secondary metrics (rank 0, N=1): the recogniser's lines/s and the default model set, in this same process
    secondary = None
    if rank == 0 and world == 1 and args.workload == "analyzer" and not (DRY or args.roofline_only or args.no_secondary):
        # a failing secondary leg must not cost the run its headline line: the error is reported in its place
        secondary = {}
        for name, fn in (("pages_per_s_exact_fp32", lambda: {"pages_per_s_exact_fp32": exact_fp32_metrics(args, an, host_pages)}),
                          ("pages_per_s_stage_control", lambda: {"pages_per_s_stage_control": stage_control_metrics(args, device, sds, pages)}),
                          ("pages_per_s_unmodified_serve", lambda: {"pages_per_s_unmodified_serve": unmodified_serve_metrics(args, device, sds, host_pages)}),
                          ("recogniser_and_default_model_set", lambda: secondary_metrics(args, device, sds, pages))):
            leg("secondary: " + name)
            try:
                secondary.update(fn())
            except Exception as exc:  # noqa: BLE001
                secondary[name] = {"error": f"{type(exc).__name__}: {exc}"}
            if name == "pages_per_s_exact_fp32":
                an.close()  # the headline's analyzer is done (its nets rebuild on demand): ~100 GB of workspaces back before the next legs reserve theirs
            torch.cuda.empty_cache()  # a finished leg's page / crop tensors go back to the device before the next one reserves

    # ---- CPU baseline leg (rank 0, N=1): oracle chain on the host cores, bounded sample
    cpu = None
    leg("cpu_baseline: oracle chain on the host cores")
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not DRY and not args.roofline_only:
        if args.workload == "analyzer":
            charset = an.text_recognizer.charset
            # one page through the CPU chain costs 13 s on an idle 128-core host and 46 s on a loaded one (both seen on the
            # gpurun boxes): one warm-up page, then timed pages until 40 s are spent - at least one, at most three
            times = cpu_timed(lambda i: cpu_analyzer_page(sds, pages[i % len(pages)], charset, args.model_set), 1, 3, 40.0)
            sample = (f"the same synthetic pages through the oracle restatement (PyTorch-CPU fp32) of the `-d cpu` chain with the "
                      f"{args.model_set} model set: detector + recogniser + layout + table nets with their pre/post-processing (PyTorch "
                      f"path for the detector too: onnxruntime is not installed); 1 warm-up page, {len(times)} timed, median")
        else:
            from oracle.dbnet import dbnet_forward

            xc = x[:1].cpu()
            times = cpu_timed(lambda i: dbnet_forward(sds["det"], xc), 1, 4, 30.0)
            sample = f"1x3x1600x1184 through oracle/dbnet.py (detector net only); 1 warm-up, {len(times)} timed, median"
        cpu = {"value": round(1.0 / float(np.median(times)), 4), "unit": "pages/s", "cores": torch.get_num_threads(), "kind": "port",
               "runs": [round(1.0 / t, 4) for t in times], "sample": sample}

    if rank == 0:
        line = {
            "metric": metric,
            "value": round(pages_job * args.steps / dt, 3) if dt else None,
            "unit": "pages/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3) if dt else None,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": SPLIT_MODES[split_mode()][1],
            "data": "synthetic",
            "config": {"workload": workload, "pages_per_step_per_gpu": len(seeds) if args.total_pages else args.pages,
                       "parallelism": (f"page-sharded x{world} GPU(s), one process per GPU: DocumentAnalyzer.serve, waves of {args.wave} "
                                       f"pages, {args.in_flight} waves in flight, 2 recogniser lanes"
                                       if args.workload == "analyzer" else f"page-sharded x{world} GPU(s), one batch of {args.pages} per forward"),
                       "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "checkpoints": "seeded synthetic (no network)", **extra},
            "roofline": roof,
            "cpu_baseline": cpu,
            "rccl": rccl,
            "per_rank": per_rank,
            "secondary": secondary,
            "highwater": hw.report() if hw is not None else None,
        }
        if args.total_pages:
            line["config"]["total_pages_per_step"] = args.total_pages
        if DRY:
            line["dry_run"] = True
            line["metric"] = "DRY RUN - orchestration rehearsal with stub page workers, not a measurement"
        print(json.dumps(line), flush=True)
    server.close()  # the analyzer, then (after a barrier: rank 0 is still in its roofline leg while the others arrive) the group


if DRY:  # stub page workers / pages / checkpoints for the CPU rehearsal (also in the spawned helper processes)
    from tests import bench_dry_stubs

    bench_dry_stubs.install(globals())

if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Benchmark of the MI355X DocumentAnalyzer hot path (contract: see the task prompt / DESIGN.md).

    python bench.py --gpus N --steps K --warmup W [--workload detector] [--batch 8]

One rank per GPU (torchrun env); a "step" is one pass of the hot path over one batch of synthetic
1600x1200 pages already resident in HBM.  Rank 0 prints ONE JSON line with the BASELINE.json
metric, the roofline of the dominant kernel (live HIP-event timing of every launch of the
implicit-GEMM convolution) and, at N=1, the CPU baseline (the oracle restatement of the reference's
PyTorch-CPU path on a bounded sample).
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

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def detector_inputs(batch, device, seed0=0):
    """Synthetic 1600x1200 BGR pages -> the tensor TextDetector.preprocess hands to the net."""
    from yomitoku_amd.utils.synth import synthetic_page

    import numpy as np

    pages = [synthetic_page(seed0 + i) for i in range(batch)]
    # text_detector.py:99-107: 1600x1200 -> resize_shortest_edge(1280,1600) -> 1600x1184, standardise.
    # (the exact INTER_AREA restatement lives in the product preprocess; the bench feeds the net seam)
    h, w = 1600, 1184
    out = torch.empty((batch, 3, h, w), dtype=torch.float32)
    mean = np.array([0.485, 0.456, 0.406], dtype=np.float32)
    std = np.array([0.229, 0.224, 0.225], dtype=np.float32)
    for i, p in enumerate(pages):
        t = torch.from_numpy(p[:, :w, :].astype(np.float32) / 255.0)  # cheap stand-in crop, same statistics
        t = (t - torch.from_numpy(mean)) / torch.from_numpy(std)
        out[i] = t.permute(2, 0, 1)
    return out.to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="detector", choices=["detector"])
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    from yomitoku_amd import _lib
    from yomitoku_amd import distributed as ydist
    from yomitoku_amd.nets import DBNet
    from yomitoku_amd.utils.synth import dbnet_state_dict

    rank, local_rank, world = ydist.init()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    lib = _lib.load()

    # weights: generated once on rank 0, broadcast over RCCL/xGMI
    sd = dbnet_state_dict(1234) if rank == 0 else None
    sd = ydist.broadcast_state_dict(sd, src=0, device=device)
    net = DBNet().load_state_dict(sd).to(device)

    x = detector_inputs(args.batch, device, seed0=1000 * rank)
    pages_per_step = args.batch

    def step():
        return net(x)["binary"]

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
    assert torch.isfinite(out).all()

    # ---- roofline leg: per-launch HIP events around the conv kernel, same steps
    roof = None
    if rank == 0:
        _lib.check(lib.ymk_prof_begin())
        psteps = min(args.steps, 3)
        for _ in range(psteps):
            step()
        torch.cuda.synchronize()
        ms, fl, ln = ctypes.c_double(), ctypes.c_double(), ctypes.c_int64()
        _lib.check(lib.ymk_prof_end(ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(ln)))
        if ms.value > 0:
            achieved = fl.value / (ms.value * 1e-3) / 1e12
            roof = {
                "bound": "mfma",
                "kernel": "conv_igemm (fp32 MFMA implicit GEMM)",
                "achieved": round(achieved, 2),
                "peak": FP32_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                "traffic": None,
                "launches_per_step": int(ln.value // psteps),
                "avg_launch_us": round(ms.value * 1e3 / max(1, ln.value), 2),
                "kernel_ms_per_step": round(ms.value / psteps, 3),
                "gflop_per_step": round(fl.value / psteps / 1e9, 1),
            }

    # ---- CPU baseline leg (rank 0, N=1): the oracle restatement on the host cores, bounded sample
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.dbnet import dbnet_forward

        xc = x[:1].cpu()
        dbnet_forward(sd, xc[:, :, :256, :256])  # warm the allocator / thread pool
        t1 = time.perf_counter()
        n_cpu = 0
        while n_cpu < 2 or (time.perf_counter() - t1 < 10.0 and n_cpu < 6):
            dbnet_forward(sd, xc)
            n_cpu += 1
        cdt = time.perf_counter() - t1
        cpu = {
            "value": round(n_cpu / cdt, 4),
            "unit": "pages/s",
            "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{n_cpu} pages of 1x3x1600x1184 through oracle/dbnet.py (PyTorch-CPU fp32 restatement of "
            "models/dbnet_plus.py), detector net only",
        }

    if rank == 0:
        total_pages = pages_per_step * args.steps * world
        line = {
            "metric": "pages/sec (TextDetector DBNet forward @1600x1200 -> 1x3x1600x1184)",
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
            "config": {
                "workload": "TextDetector DBNet (dbnet weights layout, seeded synthetic checkpoint) alone, "
                f"batch={args.batch} synthetic 1600x1200 pages per GPU (BASELINE.json configs[1])",
                "batch_per_gpu": args.batch,
                "input": "8x3x1600x1184 fp32 resident in HBM",
                "parallelism": f"page-sharded x{world}",
            },
            "roofline": roof,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

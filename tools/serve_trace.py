"""Where does a DocumentAnalyzer.serve job spend its time?  Host side: per-stage busy fractions and durations from the
pipeline's own trace.  Device side: run this under `rocprofv3 --kernel-trace` and feed the output directory to
tools/gpu_timeline.py - the job is bracketed by 0.6 s of silence on both sides so the trace can find it.

    python tools/serve_trace.py [--steps 3] [--wave 8] [--in-flight 3] [--pages 64] > trace.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402


class Sampler:
    """Every ~1 ms: which function is each thread in?  The sampler needs the GIL to run, so the gaps between its own
    samples show how long some thread kept the GIL without yielding it; the frames seen after a long gap name suspects."""

    def __init__(self):
        import threading

        self.samples, self.gaps, self.stop = {}, [], False
        self.thread = threading.Thread(target=self.run, name="ymk-sampler", daemon=True)

    def run(self):
        import threading

        names = {}
        last = time.perf_counter()
        before = {}
        while not self.stop:
            time.sleep(0.001)
            now = time.perf_counter()
            for t in threading.enumerate():
                names[t.ident] = t.name
            frames = sys._current_frames()
            where = {}
            for ident, fr in frames.items():
                name = names.get(ident, str(ident))
                if name == "ymk-sampler":
                    continue
                key = f"{os.path.basename(fr.f_code.co_filename)}:{fr.f_code.co_name}:{fr.f_lineno}"
                where[name] = key
                self.samples.setdefault(name, {}).setdefault(key, 0)
                self.samples[name][key] += 1
            if now - last > 0.004:
                moved = {k: (before.get(k), v) for k, v in where.items() if "wait" not in v or "wait" not in str(before.get(k))}
                self.gaps.append((round((now - last) * 1e3, 2), moved))
            last = now
            before = where

    def report(self):
        top = {name: sorted(c.items(), key=lambda kv: -kv[1])[:6] for name, c in self.samples.items()}
        gaps = sorted(self.gaps, key=lambda g: -g[0])
        return {"top_frames_per_thread": top, "sampler_gaps_over_4ms": len(gaps), "sampler_gap_total_ms": round(sum(g[0] for g in gaps), 1),
                "largest_sampler_gaps": gaps[:8]}


def wrap_timed(obj, name, label, sink):
    fn = getattr(obj, name)

    def inner(*a, **k):
        t = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            sink.setdefault(label, []).append(time.perf_counter() - t)

    setattr(obj, name, inner)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sample", action="store_true", help="1 ms stack sampler (GIL hold detector)")
    ap.add_argument("--fine", action="store_true", help="time the sub-steps of the recognise and layout stages")
    ap.add_argument("--gc", default="default", choices=["default", "off", "freeze"], help="cyclic collector during the job")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--wave", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=3)
    ap.add_argument("--pages", type=int, default=64)
    ap.add_argument("--unmodified", action="store_true", help="the product's own DocumentAnalyzer on the calibrated heads' own detections "
                    "(bench.py's `pages_per_s_unmodified_serve` leg: ~10 tables / ~340 cells per page) instead of the headline's hand-overs")
    ap.add_argument("--max-tables", type=int, default=0, help="TableStructureRecognizer.MAX_TABLES_PER_FORWARD for this run (0: the product's)")
    args = ap.parse_args()
    if args.max_tables > 0:
        from yomitoku_amd.table_structure_recognizer import TableStructureRecognizer

        TableStructureRecognizer.MAX_TABLES_PER_FORWARD = args.max_tables
    sys.setswitchinterval(float(os.environ.get("YMK_SWITCH_INTERVAL", 2e-4)))
    device = bench.rank_device(0)
    sds = bench.make_checkpoints("lite")
    sds = bench.calibrate_heads(sds, device, bench.Page(0, device))
    pages = bench.make_pages(list(range(args.pages)), device)
    if args.unmodified:
        from yomitoku_amd import DocumentAnalyzer
        from yomitoku_amd.utils.synth import dbnet_state_dict

        sds = dict(sds, det=dbnet_state_dict(8, out_bias=-1.5))  # as bench.unmodified_serve_metrics: a detector whose noise yields boxes
        an = DocumentAnalyzer(configs=bench.MODEL_SETS["lite"], device=str(device))
        for net, key in zip(bench.analyzer_nets(an), ("det", "rec", "lay", "tab")):
            net.load_state_dict(sds[key])
    else:
        an = bench.build_analyzer(device, sds, "lite")
        an.truth = pages
    host = [p.img for p in pages]
    an.serve(host, wave=args.wave, in_flight=args.in_flight)  # warm-up: shapes, workspaces, pinned rings
    torch.cuda.synchronize()
    time.sleep(0.6)
    fine = {}
    if args.fine:
        rec, lay, tab = an.text_recognizer, an.layout.layout_parser, an.layout.table_structure_recognizer
        for o, n, l in ((rec, "plan_pages", "rec.plan_pages [crops]"), (rec.model, "forward_groups", "rec.forward_groups"),
                        (rec, "forward_plan", "rec.forward_plan [recognize]"), (rec, "finish_plan", "rec.finish_plan [decode]"),
                        (lay, "forward_pages", "lay.forward_pages [layout]"), (lay, "pages_from_raw", "lay.pages_from_raw"),
                        (tab, "forward_tables", "tab.forward_tables"), (tab, "tables_from_raw", "tab.tables_from_raw [cells]"),
                        (an, "aggregate", "aggregate(page)"), (an.text_detector, "forward_pages", "det.forward_pages [detect]"),
                        (an.text_detector, "extract_boxes", "det.extract_boxes")):
            if type(o).__call__ is not object.__call__ and n == "__call__":
                continue  # instances look up __call__ on the type: timed through parse_pages instead
            wrap_timed(o, n, l, fine)
        from yomitoku_amd import base as ybase

        orig_init = ybase.BaseSchema.__init__

        def timed_init(self, **data):
            t = time.perf_counter()
            try:
                return orig_init(self, **data)
            finally:
                fine.setdefault("pydantic " + type(self).__name__, []).append(time.perf_counter() - t)

        ybase.BaseSchema.__init__ = timed_init
    import gc

    gc_log, gc_t = [], [0.0]

    def gc_cb(phase, info):
        if phase == "start":
            gc_t[0] = time.perf_counter()
        else:
            gc_log.append((info["generation"], time.perf_counter() - gc_t[0], info["collected"]))

    gc.callbacks.append(gc_cb)
    if args.gc == "off":
        gc.collect()
        gc.disable()
    elif args.gc == "freeze":
        gc.collect()
        gc.freeze()
    sampler = Sampler() if args.sample else None
    if sampler:
        sampler.thread.start()
    an._pipeline.trace = []
    t0 = time.perf_counter()
    res = an.serve(host * args.steps, wave=args.wave, in_flight=args.in_flight)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if sampler:
        sampler.stop = True
        sampler.thread.join()
    time.sleep(0.6)
    trace = an._pipeline.trace
    an._pipeline.trace = None
    stages = {}
    for name, seq, n, a, b in trace:
        stages.setdefault(name, []).append((a - t0, b - t0, n))
    ok = [r for r in res if not isinstance(r, BaseException)]
    out = {"workload": "unmodified serve" if args.unmodified else "headline hand-overs", "max_tables_per_forward": an.layout.table_structure_recognizer.MAX_TABLES_PER_FORWARD,
           "table_workspace_gb": round(an.layout.table_structure_recognizer.model.workspace_bytes / 2 ** 30, 2),
           "units_per_page": {"words": round(float(np.mean([len(r.words) for r in ok])), 1), "tables": round(float(np.mean([len(r.tables) for r in ok])), 2),
                              "cells": round(float(np.mean([sum(len(t.cells) for t in r.tables) for r in ok])), 1)} if ok else None,
           "pages_per_s": round(len(res) / dt, 2), "wall_s": round(dt, 4), "waves": len(stages.get("finish", [])),
           "failed": sum(isinstance(r, BaseException) for r in res), "stages": {}}
    for name, spans in stages.items():
        d = np.array([b - a for a, b, _ in spans])
        out["stages"][name] = {"busy_frac": round(float(d.sum() / dt), 3), "mean_ms": round(float(d.mean() * 1e3), 2),
                               "p90_ms": round(float(np.percentile(d, 90) * 1e3), 2), "max_ms": round(float(d.max() * 1e3), 2), "n": len(spans)}
    first = {seq: a for name, seq, n, a, b in trace if name == "stage_h2d"}  # wave latency: upload -> aggregated
    last = {seq: b for name, seq, n, a, b in trace if name == "finish"}
    lat = [last[s] - first[s] for s in last if s in first]
    out["wave_latency_ms"] = {"mean": round(float(np.mean(lat) * 1e3), 1), "max": round(float(np.max(lat) * 1e3), 1)}
    if fine:
        out["fine_ms"] = {k: {"n": len(v), "mean": round(float(np.mean(v)) * 1e3, 2), "max": round(float(np.max(v)) * 1e3, 2),
                              "total_per_wave": round(float(np.sum(v)) * 1e3 / max(1, out["waves"]), 2)} for k, v in fine.items()}
    out["gc"] = {"mode": args.gc, "collections": {str(g): {"n": sum(1 for x in gc_log if x[0] == g),
                                                         "total_ms": round(sum(x[1] for x in gc_log if x[0] == g) * 1e3, 1),
                                                         "max_ms": round(max([x[1] for x in gc_log if x[0] == g] or [0]) * 1e3, 1)} for g in (0, 1, 2)},
                 "tracked_objects": len(gc.get_objects())}
    if sampler:
        out["sampler"] = sampler.report()
    print(json.dumps(out))
    an.close()


if __name__ == "__main__":
    main()

"""8-process HOST rehearsal of the sharded job ("Next round" item 8; no GPU needed, no hardware curve claimed).

    python tools/host_rehearsal.py --ranks 1 2 4 8 --pages 512 [--gpu-ms-per-wave 145] [--wave 16]

What runs for real, per rank (one process per rank over gloo, `yomitoku_amd.distributed.ShardedServer`, exactly the product
classes): the core slice and thread budget, `DocumentAnalyzer.serve` with its nine stage threads and two recogniser lanes, and
EVERY host stage of the page path on realistic inputs - the C++ DB box extraction on 1600 x 1184 probability maps (~80 boxes per
page), the crop planner (perspective solves, bucketing, width-budget batching, forward chunking), token decode + NFKC, the
RT-DETR post-processor with its containment filters on 300-query outputs, table row / column / span filters and cell grids,
word -> cell / paragraph aggregation and reading order, pydantic schema construction, and the pickled `gather_object` of
every page's DocumentAnalyzerSchema on rank 0.

What is stubbed: the four networks and the crop kernels.  Each GPU stage takes a process-wide "device" lock and sleeps its
share of `--gpu-ms-per-wave` (the measured device time of a 16-page wave at ~110 pages/s: detector 48, crops 4, recogniser 58,
layout 20, tables 15 ms), so a rank can never exceed what its GPU would deliver, and hands on outputs of the right shape: the
probability map rendered from the page's true lines, per-line (ids, probs) of the right length, (logits, boxes) tensors that
decode to the page's true paragraphs / tables and to a 5 x 5 grid with two spans per table.

The question it answers: with N ranks on C host cores, do the host stages keep up with the GPUs - pages/s per rank at 1 / 2 /
4 / 8 ranks, process CPU seconds per page (=> cores a rank needs at 110 pages/s), and what the rank-0 gather costs."""
from __future__ import annotations

import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SHARE = {"detect": 48.0, "crops": 4.0, "recognize": 58.0, "layout": 20.0, "tables": 15.0}  # ms of a 145 ms wave of 16 pages


def make_analyzer_factory(args, pages):
    """-> make_analyzer(device, checkpoints, budget) for ShardedServer: the product's DocumentAnalyzer with the device stages
    replaced as the module doc says.  `pages`: [(img, truth_map, quads, tables, paragraphs)] shared by every source index."""
    import torch

    from yomitoku_amd import document_analyzer as da
    from yomitoku_amd import imaging, nets

    # no HIP device here: handles are never built, nothing is reserved, the crop kernels and the pyramid are not run
    nets.HipNet.to = lambda self, device: self
    nets.HipNet.reserve_once = lambda self, *a, **k: True
    imaging.build_pyramid = lambda page, levels: page
    gpu = threading.Lock()
    scale = args.gpu_ms_per_wave / sum(SHARE.values()) / 1e3

    def device(stage, n_pages):
        with gpu:  # ONE GPU per rank: its stages never overlap in this model (the real streams overlap a little)
            time.sleep(SHARE[stage] * scale * n_pages / 16.0)

    def cost(wave):
        """Pages of device time in the wave: a page marked heavy (second byte of its first pixel: a page full of tables, 35
        against 123 pages/s in bench.py's two serve legs) counts --heavy-factor times."""
        return sum(args.heavy_factor if int(img[0, 0, 1]) == 255 else 1.0 for img in wave.imgs)

    class HostStageAnalyzer(da.DocumentAnalyzer):
        def _truth(self, wave):
            return [pages[int(img[0, 0, 0]) % len(pages)] for img in wave.imgs]

        def _stage_detect(self, wave):
            device("detect", cost(wave))
            wave.maps = [t[1] for t in self._truth(wave)]

        # _stage_boxes: the product's (C++ extraction on the rendered maps)
        def _stage_crops(self, wave):
            rec = self.text_recognizer
            rec._collate_jobs = lambda jobs, flip=False, fixed_width=False: [None] * len(jobs)
            fake_pages = [torch.empty((p.shape[0], p.shape[1], 0), dtype=torch.uint8) for p in wave.pages]  # shape carriers
            wave.rec_plan = rec.plan_pages(fake_pages, [d.points for d in wave.dets])  # planner, bucketing, batching, chunking
            device("crops", cost(wave))

        def _stage_recognize(self, wave, lane=0):
            device("recognize", cost(wave))
            rng = np.random.default_rng(wave.seq)
            stats = []
            for _, plans in wave.rec_plan["jobs"]:
                n = len(plans)
                ids = rng.integers(1, 3000, size=(n, 101), dtype=np.int64).astype(np.int32)
                ids[np.arange(n), rng.integers(6, 24, size=n)] = 0  # an <eos> after 6-23 characters
                stats.append((ids, np.full((n, 101), 0.97, dtype=np.float32)))
            wave.rec_plan["stats"] = stats
            wave.rec_plan["tensors"] = None

        # _stage_decode: the product's (token decode, NFKC, un-permutation)
        def _stage_layout(self, wave):
            device("layout", cost(wave))
            lp = self.layout.layout_parser
            cat = {c: i for i, c in lp.label_mapper.items()}
            nq, nc = int(lp._cfg.RTDETRTransformerv2.num_queries), int(lp._cfg.RTDETRTransformerv2.num_classes)
            raw = []
            for page, t in zip(wave.pages, self._truth(wave)):
                h, w = int(page.shape[0]), int(page.shape[1])
                lg = np.full((1, nq, nc), -12.0, dtype=np.float32)
                bx = np.zeros((1, nq, 4), dtype=np.float32)
                units = [(b, cat["paragraphs"]) for b in t[4]] + [(b, cat["tables"]) for b in t[3]]
                for q, ((x0, y0, x1, y1), c) in enumerate(units[:nq]):
                    lg[0, q, c] = 4.0
                    bx[0, q] = ((x0 + x1) / 2 / w, (y0 + y1) / 2 / h, (x1 - x0) / w, (y1 - y0) / h)
                raw.append((lg, bx, (h, w)))
            wave.lay_raw = raw

        def _stage_tables(self, wave):
            lp, ts = self.layout.layout_parser, self.layout.table_structure_recognizer
            wave.lay_parsed = lp.pages_from_raw(wave.lay_raw)  # the product's post-processor + containment filters
            wave.lay_raw = None
            device("tables", cost(wave))
            nq, nc = int(ts._cfg.RTDETRTransformerv2.num_queries), int(ts._cfg.RTDETRTransformerv2.num_classes)
            cat = {c: i for i, c in ts.label_mapper.items()}
            raw = []
            for p, lay in enumerate(wave.lay_parsed):
                for t in lay.tables:
                    x1, y1, x2, y2 = (int(v) for v in t.box)
                    lg = np.full((1, nq, nc), -12.0, dtype=np.float32)
                    bx = np.zeros((1, nq, 4), dtype=np.float32)
                    q = 0
                    for i in range(5):  # five rows, five columns, two spans: the calibrated heads' counts
                        lg[0, q, cat["row"]] = 3.0
                        bx[0, q] = (0.5, (i + 0.5) / 5, 0.98, 0.19)
                        q += 1
                        lg[0, q, cat["col"]] = 3.0
                        bx[0, q] = ((i + 0.5) / 5, 0.5, 0.19, 0.98)
                        q += 1
                    for sx, sy in ((0.2, 0.1), (0.7, 0.5)):
                        lg[0, q, cat["span"]] = 3.0
                        bx[0, q] = (sx, sy, 0.38, 0.19)
                        q += 1
                    raw.append((p, lg, bx, {"size": (y2 - y1, x2 - x1), "offset": (x1, y1)}))
            wave.tab_raw = raw

        # _stage_cells / _stage_finish: the product's

    def make(device_, checkpoints, budget):
        an = HostStageAnalyzer(configs=args.configs, device="cuda")
        an.text_detector.post_threads = budget["box_threads"]
        an.text_detector._device = "cpu"  # PagePipeline: host-only form (no streams, no staging ring)
        return an

    return make


def rank_main(rank, world, port, args, q):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import logging
    import resource

    logging.disable(logging.WARNING)
    from bench import render_truth_map
    from yomitoku_amd import distributed as yd
    from yomitoku_amd import imaging
    from yomitoku_amd.utils.synth import synthetic_page_with_truth

    args.configs = {"ocr": {"text_detector": {"from_pretrained": False},
                            "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                                                "batch_bucketing": True, "source_downscale": True}},
                    "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}}}
    pages = []
    for i in range(8):  # eight distinct pages; a source's first pixel says which one it is
        img, quads, tables, paragraphs = synthetic_page_with_truth(1000 + i)
        img = img.copy()
        img[0, 0, 0] = i
        h, w = img.shape[:2]
        pages.append((img, render_truth_map(quads, (h, w), imaging.resize_shortest_edge_dims(h, w, 1280, 1600)), quads, tables, paragraphs))
    server = yd.ShardedServer(make_analyzer_factory(args, pages), None, backend="gloo", device="cpu")
    sources = [pages[i % len(pages)][0] for i in range(args.pages)]
    if args.heavy != "none":  # a mix of page costs: every world-th page (a whole static share) or a seeded quarter of the pages
        rng = np.random.default_rng(7)
        for i in range(args.pages):
            if (i % world == 0) if args.heavy == "one-rank" else (rng.random() < 0.25):
                sources[i] = sources[i].copy()
                sources[i][0, 0, 1] = 255
    server.serve_local(sources[: 2 * args.wave * world], wave=args.wave, in_flight=args.in_flight, assign=args.assign)  # warm-up: imports, pools, pydantic
    server.barrier()
    cpu0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    local = server.serve_local(sources, wave=args.wave, in_flight=args.in_flight, assign=args.assign)
    t_serve = time.perf_counter() - t0
    cpu1 = resource.getrusage(resource.RUSAGE_SELF)
    t1 = time.perf_counter()
    merged = server.gather(local, form=args.gather) if args.gather != "none" else (server._agree("rehearsal", None) or list(e for _, _, e in local))
    t_gather = time.perf_counter() - t1
    n_local = len(local)
    failed = sum(isinstance(e, BaseException) for _, _, e in local)
    row = {"rank": rank, "pages": n_local, "failed": failed, "serve_s": round(t_serve, 3), "pages_per_s": round(n_local / t_serve, 1),
           "cpu_s_per_page": round(((cpu1.ru_utime - cpu0.ru_utime) + (cpu1.ru_stime - cpu0.ru_stime)) / max(1, n_local), 4),
           "gather_s": round(t_gather, 3), "cores": server.cores, "budget": server.budget, "claims": server.last_run["claims"]}
    if rank == 0:
        import pickle

        ok = [e for _, _, e in local if not isinstance(e, BaseException)]
        row["gathered"] = len(merged)
        row["words_per_page"] = round(float(np.mean([len(e.words) for e in ok])), 1) if ok else None
        row["cells_per_page"] = round(float(np.mean([sum(len(t.cells) for t in e.tables) for e in ok])), 1) if ok else None
        row["pickled_kb_per_page"] = round(len(pickle.dumps(ok[0])) / 1024, 1) if ok else None
        row["json_kb_per_page"] = round(len(ok[0].model_dump_json()) / 1024, 1) if ok else None
        if ok:  # what one page costs rank 0 to rebuild, collector on / held back (64 pages in one blob, as gather_object does)
            import gc

            blob = pickle.dumps(ok[:64])
            t = time.perf_counter()
            pickle.loads(blob)
            row["unpickle_ms_per_page"] = round((time.perf_counter() - t) / len(ok[:64]) * 1e3, 2)
            gc.disable()
            t = time.perf_counter()
            pickle.loads(blob)
            row["unpickle_ms_per_page_gc_off"] = round((time.perf_counter() - t) / len(ok[:64]) * 1e3, 2)
            gc.enable()
            t = time.perf_counter()
            for e in ok[:64]:
                e.model_dump_json()
            row["dump_json_ms_per_page"] = round((time.perf_counter() - t) / len(ok[:64]) * 1e3, 2)
        if failed:
            row["first_failure"] = repr(next(e for _, _, e in local if isinstance(e, BaseException)))
    q.put(row)
    server.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4, 8])
    ap.add_argument("--pages", type=int, default=512, help="pages of the whole job (dealt round-robin to the ranks)")
    ap.add_argument("--wave", type=int, default=16)
    ap.add_argument("--in-flight", type=int, default=4)
    ap.add_argument("--gpu-ms-per-wave", type=float, default=145.0, help="device time of a 16-page wave (110 pages/s)")
    ap.add_argument("--gather", default="json", choices=["objects", "json", "none"], help="what travels to rank 0 (ShardedServer.run)")
    ap.add_argument("--assign", default="dynamic", choices=["dynamic", "static"], help="pages pulled a wave at a time (PageDealer) or dealt round-robin up front")
    ap.add_argument("--heavy", default="none", choices=["none", "one-rank", "random"],
                    help="page cost mix: every world-th page heavy (the whole static share of rank 0) or a seeded quarter of the pages")
    ap.add_argument("--heavy-factor", type=float, default=3.5, help="device time of a heavy page in light pages (123 / 35 pages/s)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import socket

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    table = []
    for world in args.ranks:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        procs = [ctx.Process(target=rank_main, args=(r, world, port, args, q)) for r in range(world)]
        for p in procs:
            p.start()
        rows = sorted((q.get(timeout=1800) for _ in procs), key=lambda r: r["rank"])
        for p in procs:
            p.join(timeout=120)
        r0 = rows[0]
        line = {"ranks": world, "host_cores": os.cpu_count(), "cores_per_rank": r0["cores"], "pages": sum(r["pages"] for r in rows),
                "failed": sum(r["failed"] for r in rows),
                "pages_per_s_per_rank_min": min(r["pages_per_s"] for r in rows), "pages_per_s_per_rank_max": max(r["pages_per_s"] for r in rows),
                "job_pages_per_s": round(sum(r["pages"] for r in rows) / (max(r["serve_s"] for r in rows) + r0["gather_s"]), 1),
                "assign": args.assign, "heavy": args.heavy, "serve_s_min": min(r["serve_s"] for r in rows), "serve_s_max": max(r["serve_s"] for r in rows),
                "pages_per_rank": [r["pages"] for r in rows], "claims_per_rank": [r["claims"] for r in rows],
                "gpu_bound_pages_per_s_per_rank": round(16e3 / args.gpu_ms_per_wave, 1),
                "cpu_s_per_page": round(float(np.mean([r["cpu_s_per_page"] for r in rows])), 4),
                "gather": args.gather, "rank0_gather_s": r0["gather_s"], "pickled_kb_per_page": r0.get("pickled_kb_per_page"), "json_kb_per_page": r0.get("json_kb_per_page"),
                "unpickle_ms_per_page": r0.get("unpickle_ms_per_page"), "unpickle_ms_per_page_gc_off": r0.get("unpickle_ms_per_page_gc_off"),
                "dump_json_ms_per_page": r0.get("dump_json_ms_per_page"),
                "words_per_page": r0.get("words_per_page"), "cells_per_page": r0.get("cells_per_page"), "budget": r0["budget"],
                "first_failure": r0.get("first_failure")}
        print(json.dumps(line), flush=True)
        table.append(line)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(table, f, indent=1)


if __name__ == "__main__":
    main()

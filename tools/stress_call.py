"""Fresh-process stress of `DocumentAnalyzer.__call__` (round-5 review item 1).

A first forward is where everything one-off happens - workspaces sized, split weight copies built, records zeroed - and
`__call__` runs two chains on two threads and two HIP streams.  A wrong result there does not show in a warm loop, so
this tool starts N fresh PROCESSES (P at a time), each of which builds the four nets with seeded weights, calls the
analyzer on the (1000, 1400) page whose seeded heads find tables (tests/test_baseline_configs_gpu.py), then calls it
AGAIN, and reports

  * the schema of the first call (compared by the parent against run 0: the first differing leaf is named),
  * a CRC of every network output of the first and of the second call, per stage (detector map, layout logits / boxes,
    table logits / boxes, recogniser logits) - the warm call is the reference the cold one must equal bit for bit,
  * the library's allocation counters (ymk_stat: allocations / workspace growth / weight copies built / stream waits
    inside a forward).

Arms are environment settings of the children (`--env KEY=VALUE`, repeatable), e.g.

  python tools/stress_call.py --runs 50 --parallel 4 --label default
  python tools/stress_call.py --runs 50 --parallel 4 --label publish_in_kernel --env YMK_DEBUG_OPTIONS=ar_publish=0
  python tools/stress_call.py --runs 50 --parallel 4 --label round5_lazy_with_hazard \
         --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NULL_MEMSET=1
  python tools/stress_call.py --runs 20 --label serialized --env AMD_SERIALIZE_KERNEL=3 --env HIP_LAUNCH_BLOCKING=1
  python tools/stress_call.py --runs 20 --label one_chain --no-concurrent

The summary (one JSON object) goes to stdout and, with --out, to a file (kept under profiles/)."""

from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

STAGES = ("det", "lay", "tab", "rec")


class _Recorder:
    """Stands where a net stands (`module.model`): calls through, keeps the outputs alive for hashing after the call."""

    def __init__(self, inner, name, log):
        object.__setattr__(self, "_inner", inner)
        object.__setattr__(self, "_name", name)
        object.__setattr__(self, "_log", log)

    def __getattr__(self, key):
        return getattr(self._inner, key)

    def __setattr__(self, key, value):
        setattr(self._inner, key, value)

    def __call__(self, *args, **kw):
        out = self._inner(*args, **kw)
        self._log.append((self._name, out))
        return out

    def forward_groups(self, batches):
        out = self._inner.forward_groups(batches)
        self._log.append((self._name, out[0]))
        return out


def _crc(value) -> int:
    import torch

    if isinstance(value, torch.Tensor):
        return zlib.crc32(value.detach().cpu().contiguous().numpy().tobytes())
    if isinstance(value, dict):
        c = 0
        for k in sorted(value):
            c = zlib.crc32(str(_crc(value[k])).encode(), c)
        return c
    if isinstance(value, (tuple, list)):
        c = 0
        for v in value:
            c = zlib.crc32(str(_crc(v)).encode(), c)
        return c
    return zlib.crc32(repr(value).encode())


def _first_difference(a, b, path=""):
    if type(a) is not type(b):
        return f"{path}: {type(a).__name__} != {type(b).__name__}"
    if isinstance(a, dict):
        if a.keys() != b.keys():
            return f"{path}: keys differ"
        for k in a:
            d = _first_difference(a[k], b[k], f"{path}.{k}")
            if d:
                return d
        return None
    if isinstance(a, list):
        if len(a) != len(b):
            return f"{path}: {len(a)} != {len(b)} entries"
        for i, (x, y) in enumerate(zip(a, b)):
            d = _first_difference(x, y, f"{path}[{i}]")
            if d:
                return d
        return None
    return None if a == b else f"{path}: {a!r} != {b!r}"


def _emit(args, report) -> int:
    """The report goes to the file the parent named (never through a pipe: a schema is larger than a pipe's buffer, and a
    parent that only reads after the child has exited would wait for ever)."""
    text = json.dumps(report, ensure_ascii=False)
    if args.report:
        with open(args.report, "w", encoding="utf-8") as f:
            f.write(text)
    else:
        sys.stdout.write(text + "\n")
    return 0


def fake_child(args) -> int:
    """No device: a report of the real shape and size (tests/test_profile_tools.py drives the parent with it).  YMK_FAKE_BAD
    names run indices (by YMK_FAKE_INDEX) whose first call 'loses its tables'."""
    index = int(os.environ.get("YMK_FAKE_INDEX", "0"))
    bad = index in {int(v) for v in os.environ.get("YMK_FAKE_BAD", "").split(",") if v}
    words = [{"points": [[i, i], [i + 9, i], [i + 9, i + 9], [i, i + 9]], "content": "w%d" % i, "direction": "horizontal",
              "det_score": 0.9, "rec_score": 0.8} for i in range(1500)]
    schema = {"paragraphs": [], "tables": [] if bad else [{"box": [0, 0, 5, 5], "n_row": 3, "n_col": 2}], "figures": [], "words": words}
    crc = {s: [1, 2] for s in STAGES}
    cold = dict(crc, tab=[7, 7]) if bad else crc
    stats = {k: 0 for k in ("allocs_in_forward", "arena_grows_in_forward", "lazy_panel_builds", "syncs_in_forward")}
    return _emit(args, {"schema": schema, "schema_crc": zlib.crc32(json.dumps(schema, sort_keys=True).encode()), "second_call_differs": None,
                        "counts": {"words": len(words), "paragraphs": 0, "tables": len(schema["tables"]), "cells": 0, "figures": 0},
                        "crc_first": cold, "crc_second": crc, "stats_first_call": stats, "stats_second_call": stats,
                        "seconds": {"first": 0.5, "second": 0.1}})


def child(args) -> int:
    if args.fake:
        return fake_child(args)
    import torch

    from yomitoku_amd import DocumentAnalyzer, _lib
    from yomitoku_amd.utils.synth import dbnet_state_dict, parseq_state_dict, synthetic_page_with_truth
    from yomitoku_amd.utils.synth_rtdetr import rtdetr_state_dict

    h, w = (int(v) for v in args.page.split("x"))
    img = synthetic_page_with_truth(3, h, w)[0]
    configs = {
        "ocr": {"text_detector": {"from_pretrained": False},
                "text_recognizer": {"model_name": "parseq-tiny-dynw-v4", "from_pretrained": False, "dynamic_width": True,
                                    "batch_bucketing": True, "source_downscale": True}},
        "layout_analyzer": {"layout_parser": {"from_pretrained": False}, "table_structure_recognizer": {"from_pretrained": False}},
    }
    an = DocumentAnalyzer(configs=configs, device="cuda:0")
    an.text_detector.model.load_state_dict(dbnet_state_dict(1234, out_bias=-2.0))
    an.text_recognizer.model.load_state_dict(parseq_state_dict(1235, eos_bias=6.0))
    an.layout.layout_parser.model.load_state_dict(rtdetr_state_dict(1240, num_classes=6, score_bias=-2.0))
    an.layout.table_structure_recognizer.model.load_state_dict(rtdetr_state_dict(1243, num_classes=3, score_bias=-1.0))
    an.concurrent_chains = not args.no_concurrent
    log = []
    modules = {"det": an.text_detector, "rec": an.text_recognizer, "lay": an.layout.layout_parser,
               "tab": an.layout.table_structure_recognizer}
    if not args.serve:  # (serve builds a second recogniser handle from the first one's type: the stand-in cannot stand there)
        for name, module in modules.items():
            module.model = _Recorder(module.model, name, log)
    if args.prewarm:  # the layout lane's first forwards before the concurrent call (bisection)
        an.layout(img)
        torch.cuda.synchronize()
        log.clear()
    if args.serve:
        return serve_child(args, an, log, _lib)
    t0 = time.perf_counter()
    first, _, _ = an(img)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    first_log, log[:] = list(log), []
    stats = {k: _lib.stat(k) for k in ("allocs_in_forward", "arena_grows_in_forward", "lazy_panel_builds", "syncs_in_forward")}
    t0 = time.perf_counter()
    second, _, _ = an(img)
    torch.cuda.synchronize()
    t_second = time.perf_counter() - t0
    second_log = list(log)
    stats_after = {k: _lib.stat(k) for k in stats}

    def by_stage(entries):
        out = {s: [] for s in STAGES}
        for name, value in entries:
            out[name].append(_crc(value))
        return out

    d1, d2 = first.model_dump(), second.model_dump()
    report = {
        "schema": d1,
        "schema_crc": zlib.crc32(json.dumps(d1, sort_keys=True, ensure_ascii=False).encode()),
        "second_call_differs": _first_difference(d1, d2),
        "counts": {"words": len(first.words), "paragraphs": len(first.paragraphs), "tables": len(first.tables),
                   "cells": sum(len(t.cells) for t in first.tables), "figures": len(first.figures)},
        "crc_first": by_stage(first_log), "crc_second": by_stage(second_log),
        "stats_first_call": stats, "stats_second_call": {k: stats_after[k] - stats[k] for k in stats},
        "seconds": {"first": round(t_first, 4), "second": round(t_second, 4)},
    }
    return _emit(args, report)


def serve_child(args, an, log, _lib) -> int:
    """--serve N: the multi-page entry point cold - N pages (the page with tables among two other sizes) through
    DocumentAnalyzer.serve in waves of 4 with 2 in flight: nine stage threads and six streams start their first forwards next to
    each other - then the same job warm.  Report: schema per page of the cold job (the parent compares processes) and cold
    against warm per page.  (No per-stage output CRCs here; and a serve process reserves ~100 GB of workspaces: --parallel 2.)"""
    import torch

    from yomitoku_amd.utils.synth import synthetic_page_with_truth

    shapes = [(1000, 1400), (1400, 1000), (1200, 1600)]
    imgs = [synthetic_page_with_truth(3 + i, *shapes[i % 3])[0] for i in range(args.serve)]
    t0 = time.perf_counter()
    cold = an.serve(imgs, wave=4, in_flight=2)
    torch.cuda.synchronize()
    t_first = time.perf_counter() - t0
    cold_log, log[:] = list(log), []
    stats = {k: _lib.stat(k) for k in ("allocs_in_forward", "arena_grows_in_forward", "lazy_panel_builds", "syncs_in_forward")}
    t0 = time.perf_counter()
    warm = an.serve(imgs, wave=4, in_flight=2)
    torch.cuda.synchronize()
    t_second = time.perf_counter() - t0
    warm_log = list(log)
    stats_after = {k: _lib.stat(k) for k in stats}
    bad = [i for i, (a, b) in enumerate(zip(cold, warm)) if isinstance(a, BaseException) or isinstance(b, BaseException)]
    if bad:
        raise RuntimeError(f"serve failed on pages {bad}: {cold[bad[0]]!r} / {warm[bad[0]]!r}")
    d1, d2 = [r.model_dump() for r in cold], [r.model_dump() for r in warm]

    def by_stage(entries):  # forwards of a stage run in wave order on one thread: the sequence per stage is comparable
        out = {s: [] for s in STAGES}
        for name, value in entries:
            out[name].append(_crc(value))
        return out

    report = {
        "schema": d1, "schema_crc": zlib.crc32(json.dumps(d1, sort_keys=True, ensure_ascii=False).encode()),
        "second_call_differs": _first_difference(d1, d2),
        "counts": {"words": sum(len(r.words) for r in cold), "paragraphs": sum(len(r.paragraphs) for r in cold), "tables": sum(len(r.tables) for r in cold),
                   "cells": sum(len(t.cells) for r in cold for t in r.tables), "figures": sum(len(r.figures) for r in cold)},
        "crc_first": by_stage(cold_log), "crc_second": by_stage(warm_log),
        "stats_first_call": stats, "stats_second_call": {k: stats_after[k] - stats[k] for k in stats},
        "seconds": {"first": round(t_first, 4), "second": round(t_second, 4)},
    }
    return _emit(args, report)


def parent(args) -> int:
    env = dict(os.environ)
    for item in args.env:
        key, _, value = item.partition("=")
        env[key] = value
    cmd = [sys.executable, os.path.abspath(__file__), "--child", "--page", args.page]
    if args.no_concurrent:
        cmd.append("--no-concurrent")
    if args.prewarm:
        cmd.append("--prewarm")
    if args.serve:
        cmd += ["--serve", str(args.serve)]
    if args.fake:
        cmd.append("--fake")
    import tempfile

    t_start = time.time()
    results, running, launched = [], [], 0
    # host threads: every child would otherwise start one OpenMP thread per core of the box for its seeded weights - with a few
    # children alive at once the oversubscribed barriers spin for minutes
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        env.setdefault(var, str(args.host_threads))
    progress = open(args.out + ".progress", "w") if args.out else None
    with tempfile.TemporaryDirectory(prefix="ymk_stress_") as tmp:
        while launched < args.runs or running:
            if args.time_budget and time.time() - t_start > args.time_budget and launched < args.runs:
                args.runs = launched  # out of time: no further children, the ones alive are waited for
            while launched < args.runs and len(running) < args.parallel:
                k = launched
                report = os.path.join(tmp, f"run{k}.json")
                log = open(os.path.join(tmp, f"run{k}.log"), "w+")
                proc = subprocess.Popen(cmd + ["--report", report], env=dict(env, YMK_FAKE_INDEX=str(k)), cwd=ROOT, stdout=log, stderr=subprocess.STDOUT)
                running.append((k, proc, report, log, time.time()))
                launched += 1
            still = []
            for k, proc, report, log, t0 in running:
                rc = proc.poll()
                if rc is None and time.time() - t0 > args.child_timeout:
                    proc.kill()
                    proc.wait()
                    rc = "timeout"
                if rc is None:
                    still.append((k, proc, report, log, t0))
                    continue
                log.seek(0)
                tail = log.read()[-2000:]
                log.close()
                try:
                    with open(report, encoding="utf-8") as f:
                        results.append((k, dict(json.load(f), process_s=round(time.time() - t0, 1))))
                except (OSError, ValueError):
                    results.append((k, {"error": f"exit {rc}", "stderr": tail, "process_s": round(time.time() - t0, 1)}))
                if progress:  # survives a parent that is killed: one line per finished child
                    r = results[-1][1]
                    progress.write(json.dumps({"run": k, "process_s": r["process_s"], "error": r.get("error"), "schema_crc": r.get("schema_crc"),
                                               "counts": r.get("counts"), "crc_first": r.get("crc_first"), "crc_second": r.get("crc_second")}) + "\n")
                    progress.flush()
            running = still
            time.sleep(0.05)
    results.sort(key=lambda r: r[0])
    ok = [(k, r) for k, r in results if "error" not in r]
    if progress:
        progress.close()
    summary = {"label": args.label, "entry_point": f"serve({args.serve} pages, wave 4, 2 in flight)" if args.serve else "__call__", "runs": args.runs, "process_s_median": sorted(r["process_s"] for _, r in results)[len(results) // 2] if results else None, "parallel": args.parallel, "page": args.page, "env": args.env,
               "no_concurrent": bool(args.no_concurrent), "prewarm": bool(args.prewarm),
               "crashed": [{"run": k, **r} for k, r in results if "error" in r], "wall_s": round(time.time() - t_start, 1)}
    if ok:
        # the reference: the most common schema among the runs whose cold call equals their own warm call (a wrong run 0 must
        # not make every other run "differ" - and where a fault hits MOST runs, the majority of all runs would be the wrong one)
        def own_fault(r):
            return bool(r["second_call_differs"]) or any(r["crc_first"][s] != r["crc_second"][s] for s in STAGES)

        tally = {}
        for k, r in ok:
            tally.setdefault(r["schema_crc"], []).append(k)
        pool = [(k, r) for k, r in ok if not own_fault(r)] or ok
        votes = {}
        for k, r in pool:
            votes[r["schema_crc"]] = votes.get(r["schema_crc"], 0) + 1
        ref_crc = max(votes, key=lambda c: votes[c])
        ref = next(r for k, r in pool if r["schema_crc"] == ref_crc)
        differing = []
        for k, r in ok:
            if r["schema_crc"] != ref_crc:
                differing.append({"run": k, "first_difference": _first_difference(ref["schema"], r["schema"]), "counts": r["counts"]})
        cold_vs_warm = {s: 0 for s in STAGES}
        cross = {s: 0 for s in STAGES}
        details = []
        for k, r in ok:
            bad = [s for s in STAGES if r["crc_first"][s] != r["crc_second"][s]]
            for s in bad:
                cold_vs_warm[s] += 1
            off = [s for s in STAGES if r["crc_second"][s] != ref["crc_second"][s]]
            for s in off:
                cross[s] += 1
            if bad or off or r["second_call_differs"]:
                details.append({"run": k, "cold_differs_from_warm": bad, "warm_differs_from_reference_run": off,
                                "second_call_differs": r["second_call_differs"], "counts": r["counts"]})
        stat_keys = list(ok[0][1]["stats_first_call"])
        summary.update({
            "completed": len(ok), "reference_counts": ref["counts"], "distinct_schemas": len(tally),
            "runs_with_a_different_schema": differing,
            "cold_output_differs_from_warm_output_by_stage": cold_vs_warm,
            "warm_output_differs_across_processes_by_stage": cross,
            "runs_with_any_difference": details,
            "stats_first_call_max": {s: max(r["stats_first_call"][s] for _, r in ok) for s in stat_keys},
            "stats_second_call_max": {s: max(r["stats_second_call"][s] for _, r in ok) for s in stat_keys},
            "first_call_s_median": sorted(r["seconds"]["first"] for _, r in ok)[len(ok) // 2],
            "second_call_s_median": sorted(r["seconds"]["second"] for _, r in ok)[len(ok) // 2],
        })
        failed = {d["run"] for d in differing} | {c["run"] for c in summary["crashed"]}
        failed |= {d["run"] for d in details if d["cold_differs_from_warm"] or d["second_call_differs"]}
        summary["failures"] = len(failed)  # distinct runs: a run that lost its tables differs in schema AND in its cold outputs
    else:
        summary["failures"] = len(results)
    text = json.dumps(summary, ensure_ascii=False, indent=1)
    print(text)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w", encoding="utf-8") as f:
            f.write(text + "\n")
    return 0 if summary["failures"] == 0 else 1


def main() -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--runs", type=int, default=50)
    ap.add_argument("--parallel", type=int, default=4, help="fresh processes alive at once (they share the one GPU)")
    ap.add_argument("--label", default="default")
    ap.add_argument("--page", default="1000x1400", help="HxW of the synthetic page (seed 3)")
    ap.add_argument("--env", action="append", default=[], help="KEY=VALUE for the children (repeatable)")
    ap.add_argument("--no-concurrent", action="store_true", help="DocumentAnalyzer.concurrent_chains = False")
    ap.add_argument("--prewarm", action="store_true", help="run the layout chain once before the measured call")
    ap.add_argument("--serve", type=int, default=0, help="N > 0: the children run DocumentAnalyzer.serve over N pages (cold, then warm) instead of __call__")
    ap.add_argument("--out", default=None)
    ap.add_argument("--time-budget", type=float, default=0.0, help="seconds after which no further child is started (0: none)")
    ap.add_argument("--host-threads", type=int, default=8, help="OMP / MKL threads per child unless the environment says otherwise")
    ap.add_argument("--child-timeout", type=float, default=180.0, help="seconds after which a child is killed and counted as crashed")
    ap.add_argument("--child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--report", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--fake", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    return child(args) if args.child else parent(args)


if __name__ == "__main__":
    sys.exit(main())

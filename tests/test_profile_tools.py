"""The scripts that turn rocprofv3 output into the files under profiles/ (tools/_rocprof_io.py, tools/roofline_crosscheck.py,
tools/pmc_aggregate.py), on synthetic traces in both formats rocprofv3 writes: CSV and the rocpd SQLite database."""
import csv
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONV = "void ymk::conv_igemm<128, 128, 4, 2, 0, 1>(ymk::ConvK)"
OTHER = "ymk::k_layernorm(float const*, int)"


def _trace(launches, tail, reps):
    """setup convs, one warm pass, `reps` timed passes, the DBNet tail: (name, start, end) with a layernorm between convs."""
    rows, t = [], 1000
    plan = [("setup", 5, 50)] + [("warm", launches, 300)] + [("timed", launches, 200)] * reps + [("tail", tail, 400)] * reps
    for _, n, dur in plan:
        for _ in range(n):
            rows.append((CONV, t, t + dur))
            t += dur + 10
            rows.append((OTHER, t, t + 20))
            t += 30
    return rows


def _write_csv(d, rows, counters=None):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "x_kernel_trace.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Dispatch_Id", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for i, (name, a, b) in enumerate(rows):
            w.writerow(["KERNEL_DISPATCH", i + 1, name, a, b])
    if counters:
        with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
            for i, (name, _, _) in enumerate(rows):
                w.writerow([i + 1, name, counters[0], counters[1]])


def _write_db(d, rows, counters=None):
    os.makedirs(d, exist_ok=True)
    con = sqlite3.connect(os.path.join(d, "x_results.db"))
    con.execute('create table kernels (name text, dispatch_id int, start int, "end" int)')
    con.executemany("insert into kernels values (?, ?, ?, ?)", [(n, i + 1, a, b) for i, (n, a, b) in enumerate(rows)])
    if counters:
        con.execute("create table counters_collection (kernel_name text, dispatch_id int, counter_name text, value real)")
        for i, (name, _, _) in enumerate(rows):  # two dimension instances per dispatch: the reader sums them
            con.executemany("insert into counters_collection values (?, ?, ?, ?)",
                            [(name, i + 1, counters[0], counters[1] / 2), (name, i + 1, counters[0], counters[1] / 2)])
    con.commit()
    con.close()


def _line(path, launches_per_page, tail_per_page, reps):
    roof = {"launches_per_page": launches_per_page, "avg_launch_us": 0.2, "kernel_ms_per_page": 0.2e-3 * launches_per_page,
            "gflop_per_page": 1.0, "achieved": 1.0, "algorithmic_bytes_per_launch": 1000,
            "serial_passes_tflops": [1.0] * reps, "dbnet_conv": {"launches_per_page": tail_per_page, "batch": 8}}
    with open(path, "w") as f:
        f.write("some log line\n" + json.dumps({"roofline": roof}) + "\n")


def test_crosscheck_picks_the_timed_passes_in_both_formats(tmp_path):
    reps, pages = 3, 16
    launches, tail = 2 * pages, 8  # launches_per_page 2, DBNet tail 1 launch per page at batch 8
    rows = _trace(launches, tail, reps)
    for fmt, writer in (("csv", _write_csv), ("db", _write_db)):
        base = tmp_path / fmt
        writer(str(base / "kt"), rows)
        writer(str(base / "fetch"), rows, ("FETCH_SIZE", 4.0))
        writer(str(base / "write"), rows, ("WRITE_SIZE", 2.0))
        _line(str(base / "line.json"), 2.0, 1.0, reps)
        out, traffic = str(base / "out.json"), str(base / "traffic.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_crosscheck.py"), str(base / "line.json"), str(base / "kt"), out,
                        str(base / "fetch"), str(base / "write"), traffic], check=True, capture_output=True)
        got = json.load(open(out))
        assert got["timed_block_launches"] == launches * reps and got["pages_in_block"] == pages * reps
        assert got["rocprof_avg_launch_us"] == 0.2  # the 200 ns launches of the timed passes, not warm (300) / tail (400)
        assert abs(got["conv_share_of_gpu_time_in_block"] - 200 / 220) < 1e-3
        t = json.load(open(traffic))
        assert t["fetch_bytes_per_launch_as_reported"] == 4096 and t["write_bytes_per_launch"] == 2048
        assert t["hbm_bytes_per_launch"] == 2 * 4096 + 2048


def test_pmc_aggregate_sums_per_kernel(tmp_path):
    rows = _trace(4, 2, 1)
    _write_db(str(tmp_path / "pmc"), rows, ("FETCH_SIZE", 3.0))
    dst = str(tmp_path / "sum.csv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_aggregate.py"), "sum", str(tmp_path / "pmc"), dst], check=True,
                   capture_output=True)
    got = {r["kernel"]: (int(r["dispatch_rows"]), float(r["sum"])) for r in csv.DictReader(open(dst))}
    n_conv = sum(1 for r in rows if r[0] == CONV)
    assert got[CONV] == (n_conv, 3.0 * n_conv) and got[OTHER][0] == len(rows) - n_conv


def test_gpu_timeline_finds_the_job_and_its_busy_fraction(tmp_path):
    """tools/gpu_timeline.py on a synthetic kernel trace: set-up dispatches, 0.6 s of silence, a job of two overlapping
    streams with one 5 ms hole, silence again.  The job is the stretch between the silences; busy = union of intervals."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gpu_timeline

    ms = 1_000_000
    rows = [(CONV, 0, 2 * ms), (OTHER, 3 * ms, 4 * ms)]  # set-up
    t0 = 700 * ms
    for i in range(10):  # stream A: 10 x 8 ms back to back; stream B: 4 ms kernels inside them
        rows.append((CONV, t0 + i * 8 * ms, t0 + (i + 1) * 8 * ms))
        rows.append((OTHER, t0 + i * 8 * ms + 2 * ms, t0 + i * 8 * ms + 6 * ms))
    rows.append((CONV, t0 + 85 * ms, t0 + 95 * ms))  # after a 5 ms hole
    rows.append((OTHER, t0 + 800 * ms, t0 + 801 * ms))  # teardown, after the closing silence
    d = str(tmp_path / "kt")
    _write_csv(d, rows)
    out = gpu_timeline.analyse(gpu_timeline.kernel_rows(d))
    assert out["dispatches"] == 21 and abs(out["wall_ms"] - 95.0) < 1e-6
    assert abs(out["gpu_busy_frac"] - 90.0 / 95.0) < 1e-3
    assert abs(out["sum_of_kernel_time_over_wall"] - 130.0 / 95.0) < 1e-2
    assert out["idle_ms_by_gap_size"][">1ms"] == 5.0 and out["largest_gaps"][0]["ms"] == 5.0
    assert set(out["kernel_time_ms_by_family"]) == {"conv_igemm", "k_layernorm"}


def test_pmc_aggregate_by_grid_tells_the_shapes_of_one_kernel_apart(tmp_path):
    d = str(tmp_path / "pmc")
    os.makedirs(d)
    with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Grid_Size", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for i in range(4):  # two layer shapes on the same instantiation: 3 dispatches of one, 1 of the other
            grid = 1024 if i < 3 else 4096
            w.writerow([i + 1, grid, CONV, "SQ_WAVES", 100 * (i + 1), 1000 * i, 1000 * i + 200 + 100 * i])
    out = str(tmp_path / "bygrid.csv")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_aggregate.py"), "bygrid", d, out], check=True, capture_output=True)
    with open(out, newline="") as f:
        rows = {r["grid"]: r for r in csv.DictReader(f)}
    assert rows["1024"]["dispatches"] == "3" and float(rows["1024"]["sum"]) == 600 and float(rows["1024"]["mean_per_dispatch"]) == 200
    assert float(rows["1024"]["mean_duration_us"]) == 0.3 and rows["4096"]["dispatches"] == "1" and float(rows["4096"]["mean_duration_us"]) == 0.5


def test_two_roof_table_prices_each_shape_against_its_binding_roof(tmp_path):
    lines = []
    def dump(i, shape, tile, ksplit, us, flop, mb):
        lines.append(f"[ymk-prof] {i:3d} {shape} tile={tile} ksplit={ksplit} grid=100  {us:8.1f} us  {flop / (us * 1e-6) / 1e12:6.1f} TFLOP/s  {mb:7.1f} MB\n")
    for span in range(3):  # three repetitions of the pass, then a foreign span that must be ignored
        # MFMA-bound on the fp16-plane kernel: 833.3 TFLOP/s-equivalent -> 83.33 GFLOP take 100 us at the roof; measured 400
        dump(0, "M=  59200 Cin= 512 Cout= 512 k=3x3 s=1 d=2 res=0", "128x128", 161, 400.0, 83.3333e9, 100.0)
        # HBM-bound: 1600 MB at 8 TB/s = 200 us; measured 500
        dump(1, "M= 947200 Cin=  64 Cout= 256 k=1x1 s=1 d=1 res=1", "128x128", 160, 500.0, 1e9, 1600.0)
        # exact fp32 kernel: 157.3 TFLOP/s -> 15.73 GFLOP take 100 us; measured 100 (on the roof)
        dump(2, "M=   4400 Cin= 256 Cout= 256 k=3x3 s=1 d=1 res=0", "64x32", 4, 100.0, 15.73e9, 1.0)
    dump(0, "M=      1 Cin=   4 Cout=   4 k=1x1 s=1 d=1 res=0", "64x64", 1, 9999.0, 1.0, 1.0)
    src, dst = tmp_path / "dump.txt", tmp_path / "out.md"
    src.write_text("noise\n" + "".join(lines))
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "two_roof.py"), str(src), str(dst), "3"], check=True, capture_output=True)
    text = dst.read_text().splitlines()
    assert text[0].startswith("3 launches per pass, 1.00 ms measured, 0.40 ms at the binding roofs (= 0.400)")
    body = [ln for ln in text if ln.startswith("| `M=")]
    assert len(body) == 3
    first = [c.strip() for c in body[0].strip("|").split("|")]    # sorted by the time above the bound: the 3 x 3 layer and the expand tie at 300 us
    by_shape = {ln.split("`")[1]: [c.strip() for c in ln.strip("|").split("|")] for ln in body}
    conv3 = by_shape["M=59200 Cin=512 Cout=512 k=3x3 s=1 d=2 res=0"]
    assert conv3[2] == "f16" and conv3[7] == "mfma" and conv3[8] == "0.25" and conv3[9] == "0.300"
    expand = by_shape["M=947200 Cin=64 Cout=256 k=1x1 s=1 d=1 res=1"]
    assert expand[7] == "hbm" and expand[8] == "0.40" and expand[6] == "3.20"
    exact = by_shape["M=4400 Cin=256 Cout=256 k=3x3 s=1 d=1 res=0"]
    assert exact[2] == "f32" and exact[8] == "1.00" and first[1] in ("128x128",)


def test_two_roof_bound_of_the_bench_line():
    sys.path.insert(0, ROOT)
    import bench

    table = [(0.4, 83.3333e9, 100e6, 3.0), (0.5, 1e9, 1600e6, 3.0), (0.1, 15.73e9, 1e6, 0.0)]
    r = bench.two_roof_bound(table)
    assert r["launches"] == 3 and r["measured_ms"] == 1.0 and abs(r["two_roof_bound_ms"] - 0.4) < 1e-3
    assert abs(r["frac_of_two_roof_bound"] - 0.4) < 1e-3 and r["hbm_bound_launches"] == 1 and r["hbm_bound_share_of_measured_time"] == 0.5
    # with the achievable HBM rate the HBM-bound launch's bound grows by 8 / 6.29
    assert abs(r["frac_with_achievable_hbm"] - (0.1 + 0.2 * 8000.0 / 6290.0 + 0.1)) < 2e-3
    assert bench.two_roof_bound([]) is None


def test_hbm_table_divides_bytes_by_kernel_time(tmp_path):
    stats, fetch, write, out = (str(tmp_path / n) for n in ("stats.csv", "fetch.csv", "write.csv", "out.md"))
    with open(stats, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs"])
        w.writerow([CONV, 10, 1_000_000, 100_000])
        w.writerow([OTHER, 5, 100, 20])  # below 0.05 % of the time: left out
    for path, counter, kb in ((fetch, "FETCH_SIZE", 1_000_000.0), (write, "WRITE_SIZE", 500_000.0)):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "counter", "dispatch_rows", "sum"])
            w.writerow([CONV, counter, 10, kb])
            w.writerow([OTHER, counter, 5, 1.0])
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "hbm_table.py"), stats, fetch, write, out], check=True, capture_output=True)
    rows = [ln for ln in open(out).read().splitlines() if ln.startswith("| `")]
    assert len(rows) == 1
    cells = [c.strip() for c in rows[0].strip("|").split("|")]
    # (2 x 1e6 KB + 0.5e6 KB) x 1024 B over 1 ms = 2.56 TB/s; 256 MB per call
    assert cells[1] == "10" and cells[4] == "256.0" and cells[5] == "2.56" and cells[6] == "0.41"


def test_stress_call_parent_names_the_runs_that_differ(tmp_path):
    """tools/stress_call.py on stand-in children (no device): reports are larger than a pipe's buffer, two of twelve
    'lose their tables' in their cold call - the parent must finish, name them and the stage whose cold output differed."""
    out = str(tmp_path / "stress.json")
    env = dict(os.environ, YMK_FAKE_BAD="3,7")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_call.py"), "--fake", "--runs", "12", "--parallel", "4", "--label", "t",
                        "--out", out], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 1, r.stderr[-500:]
    d = json.load(open(out))
    assert d["completed"] == 12 and d["distinct_schemas"] == 2 and d["reference_counts"]["tables"] == 1
    assert [x["run"] for x in d["runs_with_a_different_schema"]] == [3, 7]
    assert d["runs_with_a_different_schema"][0]["first_difference"] == ".tables: 1 != 0 entries"
    assert d["cold_output_differs_from_warm_output_by_stage"] == {"det": 0, "lay": 0, "tab": 2, "rec": 0}
    clean = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_call.py"), "--fake", "--runs", "5", "--parallel", "2"],
                           capture_output=True, text=True, timeout=120)
    assert clean.returncode == 0 and json.loads(clean.stdout)["failures"] == 0


def test_e2e_eval_counts_leaves_per_stage():
    """tools/e2e_oracle_eval.py: the leaf comparison and the verdict of a page on hand-made records (no device)."""
    import copy
    import importlib.util

    spec = importlib.util.spec_from_file_location("e2e_oracle_eval", os.path.join(ROOT, "tools", "e2e_oracle_eval.py"))
    ev = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ev)
    word = lambda i: {"points": [[i, 0], [i + 9, 0], [i + 9, 9], [i, 9]], "content": f"w{i}", "direction": "horizontal", "rec_score": 0.9, "det_score": 0.8}  # noqa: E731
    cell = {"col": 1, "row": 1, "col_span": 1, "row_span": 1, "box": [0, 0, 5, 5], "contents": "w0"}
    table = {"box": [0, 0, 50, 50], "n_row": 1, "n_col": 1, "rows": [{"box": [0, 0, 50, 10], "score": 0.7}], "cols": [{"box": [0, 0, 10, 50], "score": 0.6}],
             "spans": [], "cells": [cell], "order": 1}
    page = {"words": [word(0), word(20)], "paragraphs": [{"box": [0, 0, 9, 9], "contents": "w0", "direction": "horizontal", "order": 0, "role": None}],
            "tables": [table], "figures": []}
    same = ev.compare_pages(copy.deepcopy(page), page)
    assert all(r["differing"] == 0 for k, r in same.items() if k != "max_abs_score_diff")
    assert same["words"]["leaves"] == 2 * 10 and same["cells"]["leaves"] == 9 and same["max_abs_score_diff"] == {"det_score": 0.0, "rec_score": 0.0}
    other = copy.deepcopy(page)
    other["words"][1]["points"][0][0] += 1   # one coordinate
    other["words"][1]["rec_score"] = 0.85    # a float: not a discrete leaf
    other["tables"][0]["cells"][0]["contents"] = "x"
    other["paragraphs"].append(dict(page["paragraphs"][0], order=2))  # an element without a partner: all of its leaves
    rep = ev.compare_pages(other, page)
    assert rep["words"]["differing"] == 1 and rep["cells"]["differing"] == 1 and rep["tables"]["differing"] == 1
    assert rep["paragraphs"]["differing"] == 8 and rep["paragraphs"]["elements"] == [2, 1]
    assert abs(rep["max_abs_score_diff"]["rec_score"] - 0.05) < 1e-12
    assert ev.classify([], 2e-3) == "unexplained" and ev.classify([{"margin": 1e-4}], 2e-3) == "borderline"
    assert ev.classify([{"margin": 1e-4}, {"margin": 0.3}], 2e-3) == "failure" and ev.classify([{"margin": None}], 2e-3) == "failure"

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

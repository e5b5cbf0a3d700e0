"""Cross-check bench.py's live HIP-event roofline against rocprofv3 of the same command.

    rocprofv3 --kernel-trace --stats -d DIR -o kt -- python bench.py --roofline-only --no-cpu-baseline > line.json
    python tools/roofline_crosscheck.py line.json DIR out.json [FETCH_DIR WRITE_DIR traffic.json]

`bench.py --roofline-only` runs, in this order: the serial pass once untimed (shapes seen once), the SAME serial pass
with a HIP event pair around every conv launch (L launches: `launches_per_page` x pages of the pass; repeated, the
median pass is reported), then the DBNet forward alone (same repetitions).  The conv dispatches of the kernel trace therefore END with [L warm][L timed][DBNet]; this script takes the
timed block and compares its average duration with the live `avg_launch_us`.  With the two PMC directories (separate
`--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command) it also writes the HBM bytes per launch of the same
block, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes (KB units; FETCH_SIZE doubled on gfx950).
"""

from __future__ import annotations

import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _rocprof_io import counter_rows, kernel_rows  # noqa: E402

CONV = ("ymk::conv_igemm<", "ymk::conv_splitk<", "ymk::conv_igemm_split<", "ymk::conv_f16_dma<", "ymk::conv_f16_astat<", "ymk::k_vit_mlp_f16<")


def _is_conv(name):
    return any(tag in name for tag in CONV)


def _timed_block(rows, launches, tail):
    """The conv dispatches of the timed serial pass: the `launches` before the last `tail` (the closing DBNet forward).
    Counted from the END of the trace, because model set-up (bias calibration forwards) also launches convs."""
    conv = [r for r in rows if _is_conv(r["Kernel_Name"])]
    if len(conv) < launches + tail:
        raise SystemExit(f"{len(conv)} conv dispatches in the trace, expected at least {launches + tail}")
    return conv[len(conv) - tail - launches: len(conv) - tail], len(conv)


def _counter_block(pmc_dir, counter, launches, tail):
    rows = [r for r in counter_rows(pmc_dir) if r["Counter_Name"] == counter]
    block, total = _timed_block(rows, launches, tail)
    return sum(float(r["Counter_Value"]) for r in block) * 1024.0 / len(block), total


def block_shape(line_path):
    """(roofline dict, pages, conv launches, DBNet tail launches) of the timed block of a `--roofline-only` bench line."""
    with open(line_path) as f:
        line = json.loads([ln for ln in f.read().splitlines() if ln.startswith("{")][-1])
    roof = line["roofline"]
    pages = 16  # bench.py: prof_waves = waves[:max(1, 16 // wave)]
    if "launches_per_page" not in roof:
        raise SystemExit("not an analyzer/detector roofline line")
    launches = int(round(roof["launches_per_page"] * pages))
    reps = len(roof.get("serial_passes_tflops", [0]))  # bench.py repeats the profiled pass and reports the median one
    pages *= reps
    launches *= reps
    db = roof.get("dbnet_conv")
    tail = int(round(db["launches_per_page"] * db["batch"])) * reps if db else 0
    return roof, pages, launches, tail


def _bench_stamp():
    """What bench.py compares before it quotes these numbers: the hash of the convolution kernels' sources and the operand
    form ("conv_split") of the measured process (bench.py: kernel_source_sha / split_mode)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    return {"kernel_source_sha16": bench.kernel_source_sha(), "conv_split": bench.split_mode()}


def traffic_report(roof, launches, tail, fetch_dir, write_dir, out_path):
    fetch, n_f = _counter_block(fetch_dir, "FETCH_SIZE", launches, tail)
    write, n_w = _counter_block(write_dir, "WRITE_SIZE", launches, tail)
    traffic = {
        **_bench_stamp(),
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --roofline-only "
                  "--no-cpu-baseline; conv dispatches of the timed serial pass only",
        "kernels": "conv_igemm_split<*> + conv_f16_dma<*> + conv_f16_astat<*> + k_vit_mlp_f16<*> + conv_igemm<*> + conv_splitk<*>", "launches": launches,
        "conv_dispatches_in_fetch_pass": n_f, "conv_dispatches_in_write_pass": n_w,
        "fetch_bytes_per_launch_as_reported": round(fetch), "write_bytes_per_launch": round(write),
        "hbm_bytes_per_launch": round(2.0 * fetch + write),
        "algorithmic_bytes_per_launch": roof["algorithmic_bytes_per_launch"],
        "note": "FETCH_SIZE doubled per the guide's gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM section); "
                "WRITE_SIZE as reported (uncalibrated per the guide)",
    }
    with open(out_path, "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic))


def main(argv):
    line_path, kt_dir, out_path = argv[1:4]
    roof, pages, launches, tail = block_shape(line_path)
    rows = kernel_rows(kt_dir)
    block, total = _timed_block(rows, launches, tail)
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in block]
    by_kernel = {}
    for r, d in zip(block, dur):
        k = by_kernel.setdefault(r["Kernel_Name"], [0, 0])
        k[0] += 1
        k[1] += d
    span = int(block[-1]["End_Timestamp"]) - int(block[0]["Start_Timestamp"])
    in_span = [r for r in rows if int(block[0]["Start_Timestamp"]) <= int(r["Start_Timestamp"]) <= int(block[-1]["End_Timestamp"])]
    all_kernels_ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in in_span)
    out = {
        "command": "rocprofv3 --kernel-trace --stats -- python bench.py --roofline-only --no-cpu-baseline",
        "conv_dispatches_in_trace": total, "timed_block_launches": launches, "pages_in_block": pages,
        "rocprof_avg_launch_us": round(sum(dur) / len(dur) / 1e3, 2), "bench_avg_launch_us": roof["avg_launch_us"],
        "rocprof_conv_ms_per_page": round(sum(dur) / 1e6 / pages, 4), "bench_kernel_ms_per_page": roof["kernel_ms_per_page"],
        "rocprof_tflops": round(roof["gflop_per_page"] * pages / (sum(dur) / 1e9) / 1e3, 2), "bench_tflops": roof.get("achieved_tflops", roof["achieved"]),
        "all_kernels_ms_per_page_in_block": round(all_kernels_ns / 1e6 / pages, 4),
        "conv_share_of_gpu_time_in_block": round(sum(dur) / all_kernels_ns, 4),
        "block_span_ms_per_page": round(span / 1e6 / pages, 4),
        "absmax_pass_ms_per_page_in_block": round(sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in in_span
                                                      if "k_absmax" in r["Kernel_Name"]) / 1e6 / pages, 4),
        "per_kernel_in_block": {k: {"calls": c, "total_ms": round(t / 1e6, 3), "avg_us": round(t / c / 1e3, 2)}
                                for k, (c, t) in sorted(by_kernel.items(), key=lambda kv: -kv[1][1])},
    }
    ratio = out["rocprof_avg_launch_us"] / out["bench_avg_launch_us"]
    out["rocprof_over_bench"] = round(ratio, 4)
    with open(out_path, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "per_kernel_in_block"}))
    if len(argv) >= 7:
        traffic_report(roof, launches, tail, argv[4], argv[5], argv[6])


if __name__ == "__main__":
    if len(sys.argv) == 6 and sys.argv[1] == "--traffic-only":  # --traffic-only line.json FETCH_DIR WRITE_DIR traffic.json
        r, _, n, t = block_shape(sys.argv[2])
        traffic_report(r, n, t, sys.argv[3], sys.argv[4], sys.argv[5])
    elif len(sys.argv) not in (4, 7):
        raise SystemExit(__doc__)
    else:
        main(sys.argv)

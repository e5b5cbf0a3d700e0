"""Device timeline of the job tools/serve_trace.py ran under `rocprofv3 --kernel-trace`: how busy was the GPU inside the
job, how many kernels overlapped, where were the idle gaps.

    python tools/gpu_timeline.py ROCPROF_OUT_DIR [out.json]

The job is the stretch with the most dispatches between two silences of >= 0.3 s (set-up and warm-up come first).
Reported: wall time of the job, the fraction of it with at least one kernel running (union of the dispatch intervals),
the mean number of kernels in flight, kernel time by family, and the idle gaps (total by size, the largest with the
kernels either side)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _rocprof_io import kernel_rows  # noqa: E402

FAMILIES = ("conv_igemm", "conv_splitk", "k_parseq_dec_step", "k_flash_attn", "k_greedy_step", "k_small_attn", "k_layernorm",
            "k_deform", "k_topk", "k_det_preprocess", "k_pil_resize", "k_warp", "k_crop", "k_bilinear", "k_asf", "k_maxpool")


def family(name):
    for tag in FAMILIES:
        if tag in name:
            return tag
    return "other"


def analyse(rows, silence_ns=300_000_000):
    stretches, cur, end = [], [], None
    for r in rows:
        if end is not None and r["Start_Timestamp"] - end > silence_ns:
            stretches.append(cur)
            cur = []
        cur.append(r)
        end = r["End_Timestamp"] if end is None else max(end, r["End_Timestamp"])
    stretches.append(cur)
    job = max(stretches[1:] or stretches, key=len)
    t0, t1 = job[0]["Start_Timestamp"], max(r["End_Timestamp"] for r in job)
    wall = t1 - t0
    events = sorted([(r["Start_Timestamp"], 1) for r in job] + [(r["End_Timestamp"], -1) for r in job])
    busy = depth_area = 0
    depth, prev = 0, t0
    for t, d in events:
        if depth > 0:
            busy += t - prev
            depth_area += depth * (t - prev)
        depth += d
        prev = t
    fam = {}
    for r in job:
        f = family(r["Kernel_Name"])
        fam[f] = fam.get(f, 0) + r["End_Timestamp"] - r["Start_Timestamp"]
    gaps, cover_end, last = [], job[0]["End_Timestamp"], job[0]
    for r in job[1:]:
        if r["Start_Timestamp"] > cover_end:
            gaps.append((r["Start_Timestamp"] - cover_end, last["Kernel_Name"][:60], r["Kernel_Name"][:60]))
        if r["End_Timestamp"] > cover_end:
            cover_end, last = r["End_Timestamp"], r
    gaps.sort(reverse=True)
    hist = {"<10us": 0, "10-100us": 0, "100us-1ms": 0, ">1ms": 0}
    for g, _, _ in gaps:
        hist["<10us" if g < 10_000 else "10-100us" if g < 100_000 else "100us-1ms" if g < 1_000_000 else ">1ms"] += g
    return {"dispatches": len(job), "wall_ms": round(wall / 1e6, 2), "gpu_busy_frac": round(busy / wall, 4),
            "mean_kernels_in_flight_when_busy": round(depth_area / max(1, busy), 3),
            "sum_of_kernel_time_over_wall": round(sum(fam.values()) / wall, 3),
            "kernel_time_ms_by_family": {k: round(v / 1e6, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])},
            "idle_ms_by_gap_size": {k: round(v / 1e6, 2) for k, v in hist.items()},
            "largest_gaps": [{"ms": round(g / 1e6, 3), "after": a, "before": b} for g, a, b in gaps[:12]]}


if __name__ == "__main__":
    out = analyse(kernel_rows(sys.argv[1]))
    text = json.dumps(out, indent=1)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as f:
            f.write(text + "\n")
    print(text)

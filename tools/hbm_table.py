"""HBM GB/s per kernel of one profiled command, from three rocprofv3 passes of it:

    python tools/hbm_table.py kernel_stats.csv fetch_by_kernel.csv write_by_kernel.csv out.md

kernel_stats.csv: `rocprofv3 --kernel-trace --stats --output-format csv` (total duration per kernel); the two *_by_kernel.csv
files: `tools/pmc_aggregate.py sum` over separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of the same command (KB per
kernel).  Bytes = 2 x FETCH_SIZE + WRITE_SIZE, as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE
tallies a wide coalesced read at half its bytes; Infinity-Cache hits are counted, so a kernel re-reading what an earlier
kernel left on die can show more than the HBM could deliver).  Writes a markdown table of the kernels above 0.05 % of the
profiled GPU time, against the 6.29 TB/s the guide measured as achievable."""
import csv
import sys

ACHIEVABLE_TBS = 6.29


def by_kernel(path, counter):
    out = {}
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            if r["counter"] == counter:
                out[r["kernel"]] = (int(r["dispatch_rows"]), float(r["sum"]) * 1024.0)
    return out


def main(stats_csv, fetch_csv, write_csv, dst):
    with open(stats_csv, newline="") as f:
        stats = {r["Name"]: r for r in csv.DictReader(f)}
    fetch, write = by_kernel(fetch_csv, "FETCH_SIZE"), by_kernel(write_csv, "WRITE_SIZE")
    total_ns = sum(float(r["TotalDurationNs"]) for r in stats.values())
    rows = []
    for name, r in stats.items():
        if name not in fetch:
            continue
        dur = float(r["TotalDurationNs"])
        if dur < 5e-4 * total_ns:
            continue
        nbytes = 2.0 * fetch[name][1] + write.get(name, (0, 0.0))[1]
        rows.append((dur, name, int(r["Calls"]), nbytes))
    rows.sort(reverse=True)
    with open(dst, "w") as f:
        f.write("| kernel | calls | GPU time (ms) | share | bytes / call (MB) | TB/s | of 6.29 achievable |\n|---|---|---|---|---|---|---|\n")
        for dur, name, calls, nbytes in rows:
            tbs = nbytes / (dur * 1e-9) / 1e12
            short = name.replace("void ", "").replace("ymk::", "")
            short = short[: short.index("(")] if "(" in short else short
            f.write(f"| `{short}` | {calls} | {dur / 1e6:.1f} | {100 * dur / total_ns:.1f} % | {nbytes / calls / 1e6:.1f} | {tbs:.2f} | {tbs / ACHIEVABLE_TBS:.2f} |\n")
    print(f"{dst}: {len(rows)} kernels")


if __name__ == "__main__":
    if len(sys.argv) != 5:
        raise SystemExit(__doc__)
    main(*sys.argv[1:])

# round 6, job O: what the copyBuffer launches of the serial pass are (4.5 % of its device time): memory-copy trace of the same command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06o; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $O/mc -o mc -- python $GRAFT_REPO_ROOT/bench.py --roofline-only --no-cpu-baseline < /dev/null > $O/line.json 2> $O/err.log) || tail -5 $O/err.log
ls $O/mc | head
python - $O/mc < /dev/null <<'PY'
import csv, sys, glob, collections
d = sys.argv[1]
f = glob.glob(d + "/*memory_copy_trace.csv")
if f:
    rows = list(csv.DictReader(open(f[0])))
    print("memory copies:", len(rows), rows[0].keys() if rows else None)
    by = collections.Counter(); size = collections.Counter(); dur = collections.Counter()
    for r in rows:
        k = r.get("Direction") or r.get("Name") or "?"
        b = int(r.get("Bytes", r.get("Size", 0)) or 0)
        bucket = "<=4K" if b <= 4096 else "<=64K" if b <= 65536 else "<=1M" if b <= (1 << 20) else ">1M"
        by[(k, bucket)] += 1; size[(k, bucket)] += b; dur[(k, bucket)] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    for k in sorted(by, key=lambda k: -dur[k]): print("  ", k, by[k], "copies", round(size[k] / 1e6, 2), "MB", round(dur[k] / 1e6, 2), "ms")
k = glob.glob(d + "/*kernel_trace.csv")
if k:
    rows = [r for r in csv.DictReader(open(k[0])) if "copyBuffer" in r["Kernel_Name"] or "fillBuffer" in r["Kernel_Name"]]
    c = collections.Counter((r["Kernel_Name"][:40], r.get("Grid_Size", r.get("Grid_Size_X", "?"))) for r in rows)
    print("blit kernels:", len(rows)); print(c.most_common(12))
PY
find $O/mc -name "*kernel_trace.csv" -size +3M -delete

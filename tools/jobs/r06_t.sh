# round 6, job T: the roctx ranges of the C ABI in a marker trace (YMK_ROCTX=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06t; rm -rf $O; mkdir -p $O
(cd /tmp && YMK_ROCTX=1 timeout 200 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $O/mk -o mk -- python $GRAFT_REPO_ROOT/tools/rec_only.py rec < /dev/null > $O/out.txt 2> $O/err.log) || tail -5 $O/err.log
cat $O/out.txt | tail -2; ls $O/mk
python - $O/mk < /dev/null <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*marker*trace*.csv")
print(f)
if f:
    rows = list(csv.DictReader(open(f[0])))
    print(len(rows), "marker rows;", rows[0].keys() if rows else None)
    c = collections.Counter(r.get("Function") or r.get("Name") or "?" for r in rows)
    print(c.most_common(8))
    with open(sys.argv[1] + "/../roctx_ranges_summary.txt", "w") as out:
        out.write("YMK_ROCTX=1 rocprofv3 --marker-trace --kernel-trace -- python tools/rec_only.py rec\n")
        for k, v in c.most_common(): out.write(f"{v:6d}  {k}\n")
PY
find $O/mk -name "*kernel_trace.csv" -delete

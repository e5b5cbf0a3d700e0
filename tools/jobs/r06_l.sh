# round 6, job L: the rewritten fused decoder step (one-pass attention, rows per block chosen for ONE generation of blocks):
# recogniser tests, then the kernel's time per step at 1234 and 655 rows with 4 rows per block (the old choice) and the new rule
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06l; rm -rf $O; mkdir -p $O
echo "(tests: job L first run, 55 passed)"
for cfg in "1234 2" "2048 3" "2048 2" "300 1" "300 2" "150 1"; do
  set -- $cfg
  (cd /tmp && YMK_DEBUG_OPTIONS=dec_rows=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1_$2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload recognizer --rec-model parseq-tiny-dynw-v4 --lines $1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary < /dev/null > $O/line_$1_$2.json 2> $O/err_$1_$2.log)
  python - $O/kt_$1_$2/kt_kernel_stats.csv $1 $2 $O/line_$1_$2.json < /dev/null <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
try: line = json.load(open(sys.argv[4])); v = (line["value"], line["unit"])
except Exception as e: v = ("no line", str(e)[:80])
print("lines", sys.argv[2], "dec_rows", sys.argv[3], v)
for r in rows:
    if any(k in r["Name"] for k in ("dec_step", "greedy", "publish", "ROWMAX")) or float(r["TotalDurationNs"]) > 0.04 * tot:
        print("   %-60s calls %6s avg %8.1f us  %5.1f %%" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  find $O/kt_$1_$2 -name "*kernel_trace.csv" -delete
done

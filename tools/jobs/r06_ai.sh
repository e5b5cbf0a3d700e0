# round 6, job AI: the head with three weight slabs in flight (four LDS stages)
# per launch in a kernel trace of the recogniser leg at 1234 and 2048 lines
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06ai; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_routes_gpu.py tests/test_parseq_gpu.py tests/test_conv_astat_gpu.py -m gpu -x -q < /dev/null > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
for cfg in "1234 0" "2048 0" "655 0"; do
  set -- $cfg
  (cd /tmp && YMK_DEBUG_OPTIONS=rowmax_tile=$2 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$1_$2 -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload recognizer --rec-model parseq-tiny-dynw-v4 --lines $1 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-secondary < /dev/null > $O/line_$1_$2.json 2> $O/err_$1_$2.log)
  python - $O/kt_$1_$2/kt_kernel_stats.csv $1 $2 $O/line_$1_$2.json < /dev/null <<'PY'
import csv, sys, json
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
try: line = json.load(open(sys.argv[4])); v = (line["value"], line["unit"])
except Exception as e: v = ("no line", str(e)[:80])
print("lines", sys.argv[2], "rowmax_tile", sys.argv[3], v, "all kernels %.1f ms" % (tot / 1e6))
for r in rows:
    if any(k in r["Name"] for k in ("dec_step", "greedy", "conv_igemm_split", "false, true")):
        print("   %-70s calls %6s avg %8.1f us  %5.1f %%" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
  find $O/kt_$1_$2 -name "*kernel_trace.csv" -delete
done

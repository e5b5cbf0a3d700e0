# round 6, job F: where the table-heavy leg (the unmodified product path, ~10 tables / ~340 cells per page) spends its time:
# host stage trace with sub-step timings, then the kernel trace of the same job
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06f; rm -rf $O; mkdir -p $O
timeout 400 python tools/serve_trace.py --unmodified --fine --steps 3 --wave 16 --in-flight 4 > $O/serve_trace_unmodified_fine.json 2> $O/trace.err; echo "rc $?"; tail -2 $O/trace.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06f/serve_trace_unmodified_fine.json"))
print(d["workload"], d["units_per_page"], d["pages_per_s"], "waves", d["waves"])
for k,v in sorted(d["stages"].items(), key=lambda kv:-kv[1]["busy_frac"]): print("  stage", k, v)
for k,v in sorted(d["fine_ms"].items(), key=lambda kv:-kv[1]["total_per_wave"])[:14]: print("  fine", k, v)
PY
timeout 400 python tools/serve_trace.py --steps 3 --wave 16 --in-flight 4 > $O/serve_trace_headline.json 2>> $O/trace.err; echo "rc $?"
python -c "
import json; d=json.load(open('gpurun_out/r06f/serve_trace_headline.json')); print(d['workload'], d['units_per_page'], d['pages_per_s'])
for k,v in sorted(d['stages'].items(), key=lambda kv:-kv[1]['busy_frac']): print('  stage', k, v)"
cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o unmodified -- python $GRAFT_REPO_ROOT/tools/serve_trace.py --unmodified --steps 2 --wave 16 --in-flight 4 > $GRAFT_REPO_ROOT/$O/serve_trace_unmodified_under_rocprof.json 2>> $GRAFT_REPO_ROOT/$O/trace.err; echo "rocprof rc $?"
cd $GRAFT_REPO_ROOT; ls $O/prof/* | head; f=$(ls $O/prof/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f" | cut -c1-200
# the raw kernel trace is large: keep the stats only
find $O/prof -name "*kernel_trace.csv" -size +5M -delete

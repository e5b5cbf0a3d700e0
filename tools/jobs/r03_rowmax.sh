cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03l; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_parseq_gpu.py tests/test_pipeline_gpu.py tests/test_seq_ops_gpu.py tests/test_ops_gpu.py -m gpu -q -x 2>&1 | grep -v "INFO\|^$" | tail -6
for opt in 0 1; do
  YMK_DEBUG_OPTIONS="parseq_no_rowmax=$opt" timeout 300 python bench.py --workload recognizer --rec-model parseq-tiny-dynw-v4 --steps 5 --warmup 2 --no-cpu-baseline > $O/rec_tiny_norowmax$opt.json 2>> $O/err.log || tail -3 $O/err.log
  YMK_DEBUG_OPTIONS="parseq_no_rowmax=$opt" timeout 300 python tools/serve_trace.py --steps 3 > $O/serve_norowmax$opt.json 2>> $O/err.log || tail -3 $O/err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03l/*.json")):
    d=json.load(open(f)); print(f, d.get("value") or d.get("pages_per_s"), d.get("ms_per_step"))
PY

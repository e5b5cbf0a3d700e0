cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03q; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_serving_gpu.py -m gpu -q -x 2>&1 | grep -v "INFO\|^$" | tail -4
for lanes in 1 2 2 1; do
  YMK_REC_LANES=$lanes timeout 300 python tools/serve_trace.py --steps 8 --in-flight 4 > $O/lanes${lanes}_$RANDOM.json 2>> $O/err.log || tail -3 $O/err.log
done
YMK_REC_LANES=2 timeout 300 python tools/serve_trace.py --steps 8 --in-flight 6 > $O/lanes2_f6.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03q/*.json")):
    d=json.load(open(f)); print(f, d["pages_per_s"], {k:v["busy_frac"] for k,v in d["stages"].items() if k in ("detect","recognize","tables","layout","crops")})
PY

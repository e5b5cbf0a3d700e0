cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03r; rm -rf $O; mkdir -p $O
timeout 300 python tools/serve_trace.py --steps 8 > $O/base.json 2>> $O/err.log
YMK_SWITCH_INTERVAL=0.005 timeout 300 python tools/serve_trace.py --steps 8 > $O/switch5ms.json 2>> $O/err.log
YMK_REC_LANES=3 timeout 300 python tools/serve_trace.py --steps 8 > $O/lanes3.json 2>> $O/err.log
timeout 300 python tools/serve_trace.py --steps 8 --in-flight 6 > $O/f6.json 2>> $O/err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03r/*.json")):
    d=json.load(open(f)); print(f, d["pages_per_s"], {k:v["busy_frac"] for k,v in d["stages"].items() if k in ("detect","recognize","tables","layout","crops","finish")})
PY

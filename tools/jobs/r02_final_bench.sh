cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02f; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --procs 1 --workers 1 --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json
timeout 900 python bench.py > $O/final_default.json 2> $O/final_err.log || tail -8 $O/final_err.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r02f/final_default.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["serial_passes_tflops"], r["conv_share_of_wall"], r["dbnet_conv"]["achieved"], d["cpu_baseline"]["value"], d["cpu_baseline"]["runs"])
PY
find $O -name "*_results.db" -size +20M -delete

# full GPU test suite, smoke(), the default bench line (with roofline + CPU baseline) and the recogniser workloads
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/final_default.json 2> gpurun_out/final_err.log || tail -8 gpurun_out/final_err.log
timeout 500 python bench.py --workload recognizer --rec-model parseq-tiny-dynw-v4 --steps 3 --warmup 1 > gpurun_out/final_rec_tiny.json 2>> gpurun_out/final_err.log || tail -5 gpurun_out/final_err.log
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/final_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], d["unit"], d["ms_per_step"], r["achieved"], r["frac"], r.get("traffic"), r.get("conv_share_of_wall"), d.get("cpu_baseline"))
    except Exception as e: print(f,"ERR",e)
PY

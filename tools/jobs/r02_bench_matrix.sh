# usage: bash tools/jobs/r02_bench_matrix.sh "<procs>x<workers> ..."   (bench lines into gpurun_out/bm_*.json)
cd $GRAFT_REPO_ROOT
for pw in ${1:-2x2}; do
  p=${pw%x*}; w=${pw#*x}
  timeout 400 python bench.py --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --procs $p --workers $w > gpurun_out/bm_p${p}w${w}.json 2> gpurun_out/bm_err.log || tail -5 gpurun_out/bm_err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/bm_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("conv_share_of_wall"), r.get("wall_implied",{}).get("frac"))
    except Exception as e: print(f,"ERR",e)
PY

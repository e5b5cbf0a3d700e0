# first GPU contact of round 3: the serve() pipeline tests, then the analyzer bench over in-flight depths and queue counts
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_serving_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | tail -15
for spec in "3:8" "2:8" "4:8" "3:4" "4:16"; do
  f=${spec%:*}; q=${spec#*:}
  GPU_MAX_HW_QUEUES=$q timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --in-flight $f \
      > gpurun_out/r03a_f${f}q${q}.json 2> gpurun_out/r03a_f${f}q${q}.err || tail -5 gpurun_out/r03a_f${f}q${q}.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03a_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("conv_share_of_wall"), d["per_rank"], d["config"]["hw_queues"])
    except Exception as e: print(f,"ERR",e)
PY

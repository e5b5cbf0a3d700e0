cd $GRAFT_REPO_ROOT
for pr in "layout:-1" "ocr:-1"; do
  YMK_CHAIN_PRIORITY=$pr timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/pr_${pr%%:*}.json 2> gpurun_out/pr_err.log || tail -5 gpurun_out/pr_err.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/pr_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], d["ms_per_step"], r["achieved"], r.get("conv_share_of_wall"))
    except Exception as e: print(f,"ERR",e)
PY

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03h; rm -rf $O; mkdir -p $O
timeout 900 python tools/split_eval.py > $O/split_eval.json 2> $O/err.log || tail -8 $O/err.log
cat $O/split_eval.json
for sp in 0 2 3; do
  YMK_CONV_SPLIT=$sp timeout 400 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --in-flight 4 > $O/bench_split$sp.json 2> $O/bench_split$sp.err || tail -5 $O/bench_split$sp.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03h/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], d["ms_per_step"], r["achieved"], r["frac"], r["kernel_ms_per_page"], r["dbnet_conv"])
    except Exception as e: print(f,"ERR",e)
PY

# round 4, job B: the yardstick for the fp16-split path (exact fp32 in another summation order; three bf16 planes) on 48 whole
# pages, the bench with the split path process-wide against the exact path on the same box, default-model-set kernel stats
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04b; rm -rf $O; mkdir -p $O
echo "== pages eval f16 (3 nets) + control"; SPLIT=16 CONTROL=1 timeout 900 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16_control.json 2> $O/err0.log || tail -8 $O/err0.log
cat $O/split_eval_pages_f16_control.json
echo "== pages eval bf16x3 (3 nets)"; SPLIT=3 timeout 600 python tools/split_eval_pages.py 48 > $O/split_eval_pages_bf16x3.json 2> $O/err1.log || tail -8 $O/err1.log
cat $O/split_eval_pages_bf16x3.json
for sp in 0 16; do
  echo "== bench conv_split=$sp"
  YMK_CONV_SPLIT=$sp timeout 500 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_split$sp.json 2> $O/bench_split$sp.err || tail -5 $O/bench_split$sp.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04b/bench_split*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]; print(f, d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("kernel_ms_per_page"), r.get("dbnet_conv"))
    except Exception as e: print(f,"ERR",e)
PY
echo "== default model set under rocprofv3"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_default -o default -- python bench.py --model-set default --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_default_set.json 2> $O/err5.log || tail -5 $O/err5.log
find $O/prof_default -name "*kernel_stats.csv" -exec cp {} $O/default_set_kernel_stats.csv \;
rm -rf $O/prof_default
head -30 $O/default_set_kernel_stats.csv | cut -c1-220

# last run of the round: threaded-store kernel variants, the full GPU suite, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03z; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -x 2>&1 | tail -2
for t in 12 13 14; do
  YMK_DEBUG_OPTIONS="conv_split_tile=$t" timeout 300 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -x 2>&1 | tail -1
done
ONLY="l4 3x3|l4 1x1 1024|dec 1x1|rtdetr enc|parseq fc1" VARIANTS="0,b2t3,b2t12,b2t14,b3t2,b3t12,b3t13" REPS=5 timeout 600 python tools/conv_sweep.py > $O/sweep5.txt 2> $O/err.log || tail -5 $O/err.log
cat $O/sweep5.txt
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "INFO\|^$" | tail -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err || tail -8 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03z/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["achieved"], r["frac"], r.get("conv_share_of_wall"), r["dbnet_conv"]["frac"])
print(d["cpu_baseline"]["value"], {k:v["value"] for k,v in d["secondary"].items()})
PY

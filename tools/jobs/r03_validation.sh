# full GPU test suite, smoke(), the default bench line (roofline + secondary + CPU baseline)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03v; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | grep -v "INFO\|^$" | tail -6
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err || tail -8 $O/bench_default.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03v/bench_default.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["achieved"], r["frac"], r.get("traffic"), r.get("conv_share_of_wall"), r["dbnet_conv"])
print(d["cpu_baseline"]); print(d["rccl"], d["per_rank"]); print(json.dumps(d["secondary"], indent=1)[:1800])
PY

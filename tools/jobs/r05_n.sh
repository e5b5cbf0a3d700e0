# round 5, job N: the last tree of the round - discrete outputs of 48 whole pages (default arithmetic incl. the fused ViT MLP against
# exact fp32, with the controls), the whole GPU suite as the driver runs it, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05n; rm -rf $O; mkdir -p $O
echo "== GPU suite"
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=8 --junitxml=$O/junit.xml > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -14
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench, driver form"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05n/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"], r["per_launch"]["frac_of_two_roof_bound"], r.get("traffic"), r.get("traffic_source"))
print(d["cpu_baseline"]["value"], {k:(v.get("value", v.get("error")) if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY
echo "== 48 pages, all four nets"; ALL=1 SPLIT=16 CONTROL=1 ROUTES_CONTROL=1 timeout 400 python tools/split_eval_pages.py 48 2>/dev/null | tee $O/split_eval_pages_all_four_nets.json | cut -c1-1800

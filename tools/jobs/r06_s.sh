# round 6, job S: the last tree (pinned staging for descriptors and results, roctx switch): the whole GPU suite as the driver runs it, smoke, the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06s; rm -rf $O; mkdir -p $O
echo "== GPU suite"
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=8 --junitxml=$O/junit.xml < /dev/null > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -14
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null 2>&1 | tail -3
echo "== bench, driver form"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - < /dev/null <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/r06s/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["per_launch"]["frac_of_two_roof_bound"], r.get("traffic"), r.get("traffic_source"))
print(d["cpu_baseline"], {k:(v.get("value", v.get("error")) if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
print({k: v["seconds"] for k, v in d["highwater"]["legs"].items()})
PY

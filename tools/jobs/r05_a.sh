# round 5, job A: the whole GPU suite as the driver runs it - with the junit file, the full -q output and the high-water
# record kept -, smoke, and the driver's bench command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05a; rm -rf $O; mkdir -p $O
free -g | head -2 > $O/host.txt; nproc >> $O/host.txt
echo "== GPU suite"
YMK_HIGHWATER=$O/suite_highwater.json timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=25 --junitxml=$O/junit.xml > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -45
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench, driver form"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05a/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["achieved"], r["per_launch"]["frac_of_two_roof_bound"])
print(d["cpu_baseline"]["value"], {k:(v.get("value", v.get("error")) if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
print(json.dumps(d["highwater"]))
PY

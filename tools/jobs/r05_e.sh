# round 5, job E: the tree as it stands (A-stationary routing for K <= 192 only) - the whole GPU suite as the driver runs it
# (junit, full log, high-water record kept), smoke, the driver's bench command, and the by-layer table once more
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05e; rm -rf $O; mkdir -p $O
echo "== GPU suite"
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=15 --junitxml=$O/junit.xml > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -40
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench, driver form"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"; tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05e/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["achieved"], r["per_launch"]["frac_of_two_roof_bound"], r.get("traffic"), r.get("traffic_source"))
print(r["dbnet_conv"]["kernel_ms_per_page"], r["dbnet_conv"]["frac"], r["kernel_ms_per_page"])
print(d["cpu_baseline"]["value"], {k:(v.get("value", v.get("error")) if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
print(json.dumps(d["highwater"]))
PY
YMK_DEBUG_OPTIONS="prof_dump=1" timeout 120 python bench.py --roofline-only --no-cpu-baseline > /dev/null 2> $O/dump.txt; python tools/two_roof.py $O/dump.txt $O/two_roof.md 3 | tail -2; gzip -f $O/dump.txt

# round 4, job Q: the whole GPU suite as the driver runs it, smoke, the driver's bench command (defaults: waves of 16)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04q; rm -rf $O; mkdir -p $O
echo "== GPU suite"; timeout 1500 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "INFO\|^$" | tail -6
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench, driver form"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err || tail -8 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04q/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["achieved"], r["mfma"]["frac"], r["hbm"]["frac_of_achievable"], r.get("conv_share_of_wall"), r.get("traffic"))
print(d["cpu_baseline"]["value"], {k:(v.get("value") if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
print(d["secondary"].get("pages_per_s_unmodified_serve"))
PY

# round 4, job R: the driver's bench command once more (secondary legs timed over two passes as one job)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04r; rm -rf $O; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err || tail -8 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04r/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["achieved"], r["mfma"]["frac"], r["hbm"]["frac_of_achievable"], r.get("conv_share_of_wall"), r.get("traffic"))
print(d["cpu_baseline"]["value"], {k:(v.get("value") if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY

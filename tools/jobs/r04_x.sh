# round 4, job X: after the per-launch table went into ymk_conv.hip (the kernels themselves unchanged) - HBM traffic of the
# conv launches again (FETCH_SIZE / WRITE_SIZE, separate passes) so that the stamp matches the sources, then the driver's command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04x; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_f.json $O/fetch $O/write $O/traffic.json | cut -c1-600
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
rm -rf $O/fetch $O/write
cp $O/traffic.json profiles/r04_analyzer_pmc_conv_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err || tail -8 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04x/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["frac"], r["hbm"]["frac_of_achievable"], r.get("traffic"), r.get("traffic_source"))
print(r["per_launch"])
print(d["cpu_baseline"]["value"], {k:(v.get("value") if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY

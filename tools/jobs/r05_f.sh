# round 5, job F: discrete outputs of 48 whole pages with the round-5 arithmetic against exact fp32 (with the round-4 routing and
# an fp32 reorder as controls), the recogniser at wave scale, HBM traffic passes on the final kernel sources, then the whole suite
# and the driver's bench command once more on the final tree
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05f; rm -rf $O; mkdir -p $O
echo "== 48 pages, all four nets"; ALL=1 SPLIT=16 CONTROL=1 ROUTES_CONTROL=1 timeout 400 python tools/split_eval_pages.py 48 2>/dev/null | tee $O/split_eval_pages_all_four_nets.json | cut -c1-1500
echo "== recogniser at wave scale"; timeout 200 python tools/split_eval_parseq.py 2>/dev/null | tee $O/split_eval_parseq.json | cut -c1-900
echo "== traffic"
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_f.json $O/fetch $O/write $O/traffic.json | cut -c1-900
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
rm -rf $O/fetch $O/write
cp $O/traffic.json profiles/r05_analyzer_pmc_conv_traffic.json
echo "== GPU suite"
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=8 --junitxml=$O/junit.xml > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -14
echo "== bench, driver form"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r05f/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["achieved"], r["per_launch"]["frac_of_two_roof_bound"], r.get("traffic"), r.get("traffic_source"))
print(d["cpu_baseline"]["value"], {k:(v.get("value", v.get("error")) if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY

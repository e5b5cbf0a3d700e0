# round 6, job AF: the final tree after the row-max head moved to the A-stationary kernel - traffic passes on the final kernel sources, kernel trace cross-check, the whole GPU suite as the
# driver runs it, smoke, the driver's bench command, the by-layer table, the fresh-process stress of the final library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06af; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --no-cpu-baseline"
echo "== traffic"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B < /dev/null > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B < /dev/null > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
timeout 100 python tools/roofline_crosscheck.py --traffic-only $O/line_f.json $O/fetch $O/write $O/traffic.json < /dev/null | cut -c1-900
timeout 100 python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv < /dev/null; timeout 100 python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv < /dev/null
rm -rf $O/fetch $O/write
cp $O/traffic.json profiles/r06_analyzer_pmc_conv_traffic.json
echo "== kernel trace"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B < /dev/null > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
timeout 100 python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json < /dev/null | cut -c1-700
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
echo "== GPU suite"
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=8 --junitxml=$O/junit.xml < /dev/null > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -14
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" < /dev/null 2>&1 | tail -3
echo "== bench, driver form"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $O/bench_driver_form.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - < /dev/null <<'PY'
import json
d=json.load(open("gpurun_out/r06af/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"]["achieved"], r["per_launch"]["frac_of_two_roof_bound"], r.get("traffic"), r.get("traffic_source"))
print(r["dbnet_conv"]["kernel_ms_per_page"], r["dbnet_conv"]["frac"], r["kernel_ms_per_page"], r["launches_per_page"])
print(d["cpu_baseline"]["value"], {k:(v.get("value", v.get("error")) if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY
YMK_DEBUG_OPTIONS="prof_dump=1" timeout 120 python bench.py --roofline-only --no-cpu-baseline < /dev/null > /dev/null 2> $O/dump.txt; timeout 60 python tools/two_roof.py $O/dump.txt $O/two_roof.md 3 < /dev/null | tail -2; gzip -f $O/dump.txt
echo "== stress, final library"
timeout 200 python tools/stress_call.py --parallel 4 --child-timeout 90 --runs 400 --time-budget 90 --label final_tree --out $O/stress_final_tree.json < /dev/null > /dev/null; echo "stress rc $?"
python -c "
import json; d=json.load(open('$O/stress_final_tree.json')); print('stress', d['completed'], 'runs', d['failures'], 'failures', d['distinct_schemas'], 'schemas', d['stats_first_call_max'])" < /dev/null

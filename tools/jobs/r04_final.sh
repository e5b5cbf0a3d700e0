# round 4, final job: the whole GPU suite (serial, as the driver runs it), smoke, the driver's bench command, the evidence
# passes of the headline path (kernel trace cross-check, HBM traffic, MFMA busy, HBM GB/s per kernel), whole pages against
# the exact kernels with the final fp16 kernels, kernel stats of the default model set
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04z; rm -rf $O; mkdir -p $O
echo "== GPU suite"; timeout 1200 python -m pytest tests/ -x -q -m gpu 2>&1 | grep -v "INFO\|^$" | tail -6
echo "== smoke"; timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
echo "== bench, driver form"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err || tail -8 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04z/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["bound"], r["achieved"], r["frac"], r["mfma"], r["hbm"]["frac_of_achievable"], r.get("conv_share_of_wall"), r.get("traffic"))
print(d["cpu_baseline"]["value"], {k:(v.get("value") if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json | cut -c1-700
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_kt.json $O/fetch $O/write $O/traffic.json | cut -c1-900
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
python tools/hbm_table.py $O/kernel_stats.csv $O/fetch_by_kernel.csv $O/write_by_kernel.csv $O/hbm_gbs_by_kernel.md
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -o m -- $B > $O/line_m.json 2> $O/m.log || tail -5 $O/m.log
python tools/pmc_aggregate.py sum $O/mfma $O/mfma_by_kernel.csv
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
rm -rf $O/kt $O/fetch $O/write $O/mfma
echo "== whole pages vs exact fp32 (final kernels)"
SPLIT=16 CONTROL=1 timeout 900 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16_control.json 2> $O/err_pages.log || tail -8 $O/err_pages.log
cat $O/split_eval_pages_f16_control.json
SPLIT=16 ALL=1 timeout 900 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16_all.json 2> $O/err_pages2.log || tail -8 $O/err_pages2.log
cat $O/split_eval_pages_f16_all.json
echo "== default model set under rocprofv3"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o default -- python bench.py --model-set default --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_default_set.json 2> $O/err5.log || tail -5 $O/err5.log
find $O/prof_default -name "*kernel_stats.csv" -exec cp {} $O/default_set_kernel_stats.csv \; ; rm -rf $O/prof_default
head -12 $O/default_set_kernel_stats.csv | cut -c1-160
du -sh $O

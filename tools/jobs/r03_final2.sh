# after the swizzled-tile change: the full GPU suite (4 xdist workers), smoke, the driver's bench command, kernel trace and
# MFMA-busy pass of the serial roofline pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03y; rm -rf $O; mkdir -p $O
timeout 480 python -m pytest tests -m gpu -q -x -n 4 2>&1 | grep -v "INFO\|^$" | tail -5
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form.json 2> $O/bench.err || tail -8 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03y/bench_driver_form.json")); r=d["roofline"]
print(d["value"], d["unit"], d["ms_per_step"], r["achieved"], r["frac"], r.get("conv_share_of_wall"), r["dbnet_conv"]["frac"])
print(d["cpu_baseline"]["value"], {k:(v.get("value") if isinstance(v, dict) else v) for k,v in d["secondary"].items()})
PY
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/kt
head -8 $O/kernel_stats.csv
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/mfma -o m -- $B > $O/line_m.json 2> $O/m.log || tail -5 $O/m.log
python tools/pmc_aggregate.py sum $O/mfma $O/mfma_by_kernel.csv
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/mfma
head -12 $O/mfma_by_kernel.csv

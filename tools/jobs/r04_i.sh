# round 4, job I: channel-major K order of the fp16 panels + the LDS-DMA form routed automatically: parity, sweep, bench,
# HBM traffic passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04i; rm -rf $O; mkdir -p $O
echo "== tests"; timeout 900 python -m pytest tests/test_conv_split_gpu.py tests/test_dbnet_gpu.py tests/test_rtdetr_gpu.py -m gpu -q 2>&1 | grep -v "INFO\|^$" | tail -8
echo "== sweep"; VARIANTS="b16t3,b16t21,b16" REPS=5 timeout 400 python tools/conv_sweep.py > $O/sweep.txt 2> $O/err0.log || tail -5 $O/err0.log
cut -c1-170 $O/sweep.txt
echo "== bench"
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04i/bench.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("kernel_ms_per_page"), r.get("dbnet_conv"))
PY
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_f.json $O/fetch $O/write $O/traffic.json
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
rm -rf $O/fetch $O/write

# round 4, job F: the widened split dispatch (128 x 64 tiles for half-filled launches, the greedy loop's row-max head, LDS-DMA
# form on the 3 x 3 / long-K layers): targeted parity tests, then the bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04f; rm -rf $O; mkdir -p $O
echo "== tests"; timeout 900 python -m pytest tests/test_conv_split_gpu.py tests/test_parseq_gpu.py tests/test_rtdetr_gpu.py tests/test_dbnet_gpu.py -m gpu -q -x 2>&1 | grep -v "INFO\|^$" | tail -8
echo "== bench"
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04f/bench.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("kernel_ms_per_page"), r.get("dbnet_conv"))
PY
echo "== recogniser lines/s (tiny)"
timeout 300 python bench.py --workload recognizer --rec-model parseq-tiny-dynw-v4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rec_tiny.json 2> $O/rec.err || tail -5 $O/rec.err
python -c "
import json; d=json.load(open('gpurun_out/r04f/bench_rec_tiny.json')); print(d['value'], d['unit'], d['roofline']['achieved'])"

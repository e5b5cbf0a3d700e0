# round 4, job D: the whole GPU suite with the fp16 split as the models' default and the producers' max|x| records,
# then the bench (headline form) and the serial-pass launch dump
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04d; rm -rf $O; mkdir -p $O
echo "== record self-check + split tests"
timeout 600 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -x -s 2>&1 | grep -v "INFO\|^$" | tail -8
echo "== whole GPU suite, 4 workers"
timeout 1500 python -m pytest tests/ -m gpu -q -n 4 2>&1 | grep -v "INFO\|^$" | tail -25
echo "== bench (default = fp16 split)"
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench_f16_records.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04d/bench_f16_records.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("kernel_ms_per_page"), r.get("dbnet_conv"))
PY

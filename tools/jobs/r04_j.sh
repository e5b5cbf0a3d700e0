# round 4, job J: fp16-split flash attention: operator tests, PARSeq parity, recogniser and analyzer benches
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04j; rm -rf $O; mkdir -p $O
echo "== attention tests"; timeout 600 python -m pytest tests/test_seq_ops_gpu.py -m gpu -q -s -k "flash" 2>&1 | grep -v "INFO\|^$" | tail -30
echo "== parseq tests"; timeout 900 python -m pytest tests/test_parseq_gpu.py -m gpu -q 2>&1 | grep -v "INFO\|^$" | tail -5
echo "== recogniser lines/s (tiny)"
timeout 300 python bench.py --workload recognizer --rec-model parseq-tiny-dynw-v4 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_rec_tiny.json 2> $O/rec.err || tail -5 $O/rec.err
python -c "
import json; d=json.load(open('gpurun_out/r04j/bench_rec_tiny.json')); print(d['value'], d['unit'])"
echo "== bench"
timeout 600 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err || tail -5 $O/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04j/bench.json")); r=d["roofline"]
print(d["value"], d["ms_per_step"], r["achieved"], r["frac"], r.get("kernel_ms_per_page"))
PY
echo "== default model set"
timeout 500 python bench.py --model-set default --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_default_set.json 2> $O/err5.log || tail -5 $O/err5.log
python -c "
import json; d=json.load(open('gpurun_out/r04j/bench_default_set.json')); print(d['value'], d['unit'])"

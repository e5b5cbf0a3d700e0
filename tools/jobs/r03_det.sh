cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03s; rm -rf $O; mkdir -p $O
timeout 400 python bench.py --workload detector --steps 3 --warmup 1 > $O/bench_detector.json 2> $O/err.log || tail -8 $O/err.log
python -c "
import json; d=json.load(open('gpurun_out/r03s/bench_detector.json')); r=d['roofline']; print(d['value'], d['unit'], r['achieved'], r['frac'], d['cpu_baseline'])"

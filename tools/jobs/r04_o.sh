# round 4, job O: balanced forward chunks, up to 2048 lines per grouped forward: wave size of the serve pipeline again
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04o; rm -rf $O; mkdir -p $O
echo "== tests"; timeout 900 python -m pytest tests/test_serving_gpu.py tests/test_pipeline_gpu.py -m gpu -q 2>&1 | grep -v "INFO\|^$" | tail -5
for cfg in "8 4" "16 3" "16 2" "12 4"; do
  set -- $cfg
  timeout 300 python bench.py --steps 6 --warmup 2 --wave $1 --in-flight $2 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_w$1_f$2.json 2> $O/err_w$1_f$2.log || tail -3 $O/err_w$1_f$2.log
  python -c "
import json; d=json.load(open('$O/bench_w$1_f$2.json')); print('wave $1 in_flight $2:', d['value'], d['unit'])"
done

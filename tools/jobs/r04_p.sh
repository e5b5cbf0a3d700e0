# round 4, job P: larger waves (balanced forward chunks): wave x in-flight
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04p; rm -rf $O; mkdir -p $O
for cfg in "16 4" "16 6" "24 3" "32 2" "32 3"; do
  set -- $cfg
  timeout 300 python bench.py --steps 6 --warmup 2 --wave $1 --in-flight $2 --no-cpu-baseline --no-secondary --no-roofline > $O/bench_w$1_f$2.json 2> $O/err_w$1_f$2.log || tail -3 $O/err_w$1_f$2.log
  python -c "
import json; d=json.load(open('$O/bench_w$1_f$2.json')); print('wave $1 in_flight $2:', d['value'], d['unit'])"
done

# round 6, job I: the table-heavy leg with forwards of 64 table crops: waves in flight and wave size; one RT-DETRv2 forward alone at batch 64
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06i; rm -rf $O; mkdir -p $O
timeout 120 python tools/rtdetr_profile.py --batch 64 < /dev/null 2> /dev/null | tail -1
for cfg in "16 4" "16 2" "16 1" "32 2" "8 4"; do
  set -- $cfg
  timeout 200 python tools/serve_trace.py --unmodified --steps 3 --wave $1 --in-flight $2 --max-tables 64 < /dev/null > $O/trace_w$1_f$2.json 2> $O/err.log || tail -3 $O/err.log
  python -c "
import json; d=json.load(open('$O/trace_w$1_f$2.json')); print('wave $1 in flight $2:', d['pages_per_s'], {k: (v['busy_frac'], v['mean_ms']) for k,v in d['stages'].items() if k in ('tables','cells','detect','layout','recognize')})" < /dev/null
done

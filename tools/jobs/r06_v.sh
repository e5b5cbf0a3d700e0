# round 6, job V: the one page of job U that differed in exact-fp32 mode, again with the tool pairing detections before it compares them
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06v; rm -rf $O; mkdir -p $O
timeout 300 python tools/e2e_oracle_eval.py --pages 1 --first-seed 131 --lay-seed 1248 --out $O/e2e_page_131.json < /dev/null 2> $O/err.log | cut -c1-500; grep "^page" $O/err.log
python -c "
import json; d=json.load(open('$O/e2e_page_131.json'))
for p in d['per_page']:
    for m in ('split','exact'):
        if 'roots' in p[m]: print(p['page'], m, p[m]['verdict'], json.dumps(p[m]['stages']), json.dumps(p[m]['roots'])[:1500])
" < /dev/null

# round 6, job Q: the one page that differed in exact-fp32 mode with the second layout head, with the query-selection root in the tool
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06q; rm -rf $O; mkdir -p $O
timeout 300 python tools/e2e_oracle_eval.py --pages 2 --lay-seed 1248 --out $O/e2e_lay_seed_1248_2_pages.json < /dev/null 2> $O/err.log | cut -c1-600; grep "^page" $O/err.log
python -c "
import json; d=json.load(open('$O/e2e_lay_seed_1248_2_pages.json'))
for p in d['per_page']:
    for m in ('split','exact'):
        if 'roots' in p[m]: print(p['page'], m, p[m]['verdict'], json.dumps(p[m]['roots'])[:1500])
" < /dev/null
timeout 300 python -m pytest tests/test_imaging_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu < /dev/null 2>&1 | grep -v INFO | tail -3

# round 6, job AA: the analyzer's aggregation options end to end against the free-running oracle (both layout heads)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06aa; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 400 python tools/e2e_oracle_eval.py "$@" --out $O/e2e_$name.json < /dev/null 2> $O/$name.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['pages'], d['totals'], 'failures', d['failures'], {m: (sum(v['leaves'] for v in r['stages'].values()), sum(v['differing'] for v in r['stages'].values()), r['verdicts']) for m, r in d['modes'].items()})"; grep "^page" $O/$name.err | grep -v "equal exact: equal" | head -4; }
run ignore_meta_ignore_ruby_12_pages --pages 12 --ignore-meta --ignore-ruby
run reading_order_right2left_second_head_12_pages --pages 12 --lay-seed 1248 --reading-order right2left
run reading_order_left2right_ignore_ruby_second_head_8_pages --pages 8 --first-seed 30 --lay-seed 1248 --reading-order left2right --ignore-ruby --split-text-across-cells

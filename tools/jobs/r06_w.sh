# round 6, job W: evidence at volume on the last tree - 64 pages at BASELINE's page size (both orientations) against the free-running
# oracle, and the __call__ stress for 400 s
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06w; rm -rf $O; mkdir -p $O
timeout 900 python tools/e2e_oracle_eval.py --pages 64 --first-seed 200 --shapes 1600x1200,1200x1600 --out $O/e2e_64_pages_1600x1200.json < /dev/null 2> $O/e2e.err | cut -c1-1100; echo "rc ${PIPESTATUS[0]}"; grep "^page" $O/e2e.err | grep -v "equal exact: equal" | head
timeout 560 python tools/stress_call.py --parallel 4 --child-timeout 90 --runs 2000 --time-budget 400 --label call_last_tree_400s --out $O/stress_call_last_tree_400s.json < /dev/null > /dev/null; echo "stress rc $?"
python -c "
import json; d=json.load(open('$O/stress_call_last_tree_400s.json')); print('stress', d['completed'], 'runs', d['failures'], 'failures', d['distinct_schemas'], 'schemas', d['stats_first_call_max'], d['cold_output_differs_from_warm_output_by_stage'])" < /dev/null

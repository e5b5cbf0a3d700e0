# round 6, job AJ: the final library (the greedy loop's head on the A-stationary kernel) end to end against the free-running oracle on
# pages it has not seen, and the fresh-process stress of the multi-page entry point
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06aj; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 500 python tools/e2e_oracle_eval.py "$@" --out $O/e2e_$name.json < /dev/null 2> $O/$name.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$name', d['pages'], d['totals'], 'failures', d['failures'], {m: (sum(v['leaves'] for v in r['stages'].values()), sum(v['differing'] for v in r['stages'].values()), r['verdicts']) for m, r in d['modes'].items()})"; grep "^page" $O/$name.err | grep -v "equal exact: equal" | head -4; }
run final_library_24_pages_seeds_400_423 --pages 24 --first-seed 400
run final_library_second_head_16_pages_seeds_500_515 --pages 16 --first-seed 500 --lay-seed 1248 --split-text-across-cells
python -c "
from yomitoku_amd import _lib; _lib.load(); print('astat', _lib.stat('astat_launches'))" < /dev/null 2>&1 | tail -1
S="python tools/stress_call.py --child-timeout 120 --runs 1000"
timeout 250 $S --parallel 2 --serve 12 --time-budget 150 --label serve_final_library --out $O/stress_serve_final_library.json < /dev/null > /dev/null; echo "rc $?"
python - < /dev/null <<'PY'
import json, glob
for p in sorted(glob.glob("/root/repo/gpurun_out/r06aj/stress_*.json")):
    d = json.load(open(p))
    print(d["label"], d.get("entry_point"), "completed", d.get("completed"), "failures", d["failures"], "distinct", d.get("distinct_schemas"), "cold!=warm", d.get("cold_output_differs_from_warm_output_by_stage"),
          "stats", d.get("stats_first_call_max"), "crashed", len(d["crashed"]), "proc", d.get("process_s_median"))
PY

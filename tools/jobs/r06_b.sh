# round 6, job B: fresh-process stress of DocumentAnalyzer.__call__ in five arms (tools/stress_call.py; job A lost its arms to a
# pipe deadlock in the tool's first form)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06b; rm -rf $O; mkdir -p $O
echo "== one child, by hand"
timeout 200 python tools/stress_call.py --child --report $O/one_child.json 2>&1 | grep -v INFO | tail -5
python -c "import json; d=json.load(open('$O/one_child.json')); d.pop('schema'); print(d)"
S="python tools/stress_call.py --parallel 4 --child-timeout 150"
arm() { label=$1; runs=$2; shift 2; timeout 600 $S --runs $runs --label $label "$@" --out $O/stress_$label.json > /dev/null; echo "$label rc $?"; }
arm default 50
arm publish_in_kernel 30 --env YMK_DEBUG_OPTIONS=ar_publish=0
arm lazy 30 --env YMK_DEBUG_LAZY_SPLIT=1
arm lazy_hazard_null_memset 50 --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NULL_MEMSET=1
arm lazy_hazard_no_finalize_sync 30 --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NO_FINALIZE_SYNC=1
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r06b/stress_*.json")):
    d = json.load(open(p))
    print(d["label"], "completed", d.get("completed"), "failures", d["failures"], "distinct", d.get("distinct_schemas"), "cold!=warm", d.get("cold_output_differs_from_warm_output_by_stage"),
          "cross", d.get("warm_output_differs_across_processes_by_stage"), "stats", d.get("stats_first_call_max"), "crashed", len(d["crashed"]), "wall", d["wall_s"], "first", d.get("first_call_s_median"))
    for r in d.get("runs_with_a_different_schema", [])[:5]: print("   ", r)
    for c in d["crashed"][:2]: print("   crash", c["error"], c["stderr"][-400:])
PY

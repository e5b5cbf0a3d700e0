# round 6, job A: the allocation-free forwards (split copies at finalize, ensure_workspace) on the device, then the fresh-process
# stress of DocumentAnalyzer.__call__ in five arms (tools/stress_call.py)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06a; rm -rf $O; mkdir -p $O
echo "== tests that touch what changed"
timeout 900 python -m pytest tests/test_serving_gpu.py tests/test_parseq_gpu.py tests/test_routes_gpu.py tests/test_conv_astat_gpu.py "tests/test_baseline_configs_gpu.py::test_whole_page_schema_vs_oracle_chain" -x -q -m gpu --durations=5 > $O/pytest_subset.log 2>&1
echo "subset rc $?"; grep -v "INFO\|^$" $O/pytest_subset.log | tail -12
S="timeout 900 python tools/stress_call.py --parallel 4"
echo "== stress: default tree"
$S --runs 50 --label default --out $O/stress_default.json > /dev/null; echo "rc $?"
echo "== stress: publication inside the greedy kernel (rounds 1-5)"
$S --runs 30 --label publish_in_kernel --env YMK_DEBUG_OPTIONS=ar_publish=0 --out $O/stress_publish_in_kernel.json > /dev/null; echo "rc $?"
echo "== stress: round-5 laziness, hazards closed"
$S --runs 30 --label lazy --env YMK_DEBUG_LAZY_SPLIT=1 --out $O/stress_lazy.json > /dev/null; echo "rc $?"
echo "== stress: round-5 laziness + null-stream memset of the max|x| words"
$S --runs 50 --label lazy_hazard_null_memset --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NULL_MEMSET=1 --out $O/stress_lazy_hazard_null_memset.json > /dev/null; echo "rc $?"
echo "== stress: round-5 laziness + no device synchronisation at finalize"
$S --runs 30 --label lazy_hazard_no_finalize_sync --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NO_FINALIZE_SYNC=1 --out $O/stress_lazy_hazard_no_finalize_sync.json > /dev/null; echo "rc $?"
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r06a/stress_*.json")):
    d = json.load(open(p))
    print(d["label"], "completed", d.get("completed"), "failures", d["failures"], "distinct", d.get("distinct_schemas"), "cold!=warm", d.get("cold_output_differs_from_warm_output_by_stage"),
          "cross", d.get("warm_output_differs_across_processes_by_stage"), "stats", d.get("stats_first_call_max"), "crashed", len(d["crashed"]), "wall", d["wall_s"], "first", d.get("first_call_s_median"))
    for r in d.get("runs_with_a_different_schema", [])[:5]: print("   ", r)
    for c in d["crashed"][:2]: print("   crash", c["error"], c["stderr"][-400:])
PY

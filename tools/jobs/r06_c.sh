# round 6, job C: fresh-process stress of DocumentAnalyzer.__call__ (tools/stress_call.py), every step bounded in time: a probe
# (how long a child takes alone and four at a time), then five arms of ~140 s each
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06c; rm -rf $O; mkdir -p $O
nproc; python -c "import torch; print('torch threads', torch.get_num_threads())"
S="python tools/stress_call.py --child-timeout 90"
echo "== probe: one at a time"
timeout 200 $S --runs 3 --parallel 1 --time-budget 60 --label probe1 --out $O/probe1.json > /dev/null; echo "rc $?"; cat $O/probe1.json.progress | cut -c1-120
echo "== probe: four at a time"
timeout 250 $S --runs 8 --parallel 4 --time-budget 90 --label probe4 --out $O/probe4.json > /dev/null; echo "rc $?"; cat $O/probe4.json.progress | cut -c1-120
MED=$(python -c "import json; print(int(json.load(open('$O/probe4.json')).get('process_s_median') or 999))" 2>/dev/null || echo 999)
echo "median child seconds at four a time: $MED"
if [ "$MED" -gt 45 ]; then echo "children too slow: stopping here"; exit 0; fi
arm() { label=$1; shift; timeout 260 $S --parallel 4 --runs 400 --time-budget 140 --label $label "$@" --out $O/stress_$label.json > /dev/null; echo "$label rc $?"; }
arm default
arm lazy_hazard_null_memset --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NULL_MEMSET=1
arm lazy --env YMK_DEBUG_LAZY_SPLIT=1
arm publish_in_kernel --env YMK_DEBUG_OPTIONS=ar_publish=0
arm lazy_hazard_no_finalize_sync --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NO_FINALIZE_SYNC=1
python - <<'PY'
import json, glob
for p in sorted(glob.glob("gpurun_out/r06c/stress_*.json")):
    d = json.load(open(p))
    print(d["label"], "completed", d.get("completed"), "failures", d["failures"], "distinct", d.get("distinct_schemas"), "cold!=warm", d.get("cold_output_differs_from_warm_output_by_stage"),
          "cross", d.get("warm_output_differs_across_processes_by_stage"), "stats", d.get("stats_first_call_max"), "crashed", len(d["crashed"]), "wall", d["wall_s"], "proc", d.get("process_s_median"))
    for r in d.get("runs_with_a_different_schema", [])[:5]: print("   ", r)
    for c in d["crashed"][:2]: print("   crash", c["error"], c["stderr"][-400:])
PY

# round 6, job R (second run: serve children two at a time - four reserve 400 GB of workspaces - and without the output recorders): fresh-process stress of the multi-page entry point (serve, cold then warm) on the final library, the same with the
# round-5 hazard put back, and a longer run of the __call__ arm
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06r; rm -rf $O; mkdir -p $O
S="python tools/stress_call.py --child-timeout 120 --runs 1000"
echo "== one serve child by hand"
timeout 200 python tools/stress_call.py --child --serve 12 --report $O/one_serve_child.json < /dev/null 2>&1 | grep -v INFO | tail -3
python -c "import json; d=json.load(open('$O/one_serve_child.json')); d.pop('schema'); print(d)" < /dev/null | cut -c1-900
timeout 330 $S --parallel 2 --serve 12 --time-budget 200 --label serve_final_tree --out $O/stress_serve_final_tree.json < /dev/null > /dev/null; echo "rc $?"
timeout 250 $S --parallel 2 --serve 12 --time-budget 120 --label serve_lazy_hazard_null_memset --env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NULL_MEMSET=1 --out $O/stress_serve_lazy_hazard_null_memset.json < /dev/null > /dev/null; echo "rc $?"
echo "(the __call__ arm ran in the first run of this job: 174 processes, 0 failures)"
python - < /dev/null <<'PY'
import json, glob
for p in sorted(glob.glob("/root/repo/gpurun_out/r06r/stress_*.json")):
    d = json.load(open(p))
    print(d["label"], d.get("entry_point"), "completed", d.get("completed"), "failures", d["failures"], "distinct", d.get("distinct_schemas"), "cold!=warm", d.get("cold_output_differs_from_warm_output_by_stage"),
          "stats", d.get("stats_first_call_max"), "crashed", len(d["crashed"]), "proc", d.get("process_s_median"), d.get("reference_counts"))
    for r in d.get("runs_with_a_different_schema", [])[:3]: print("   ", r)
    for c in d["crashed"][:2]: print("   crash", c["error"], c["stderr"][-300:])
PY

# round 6, job Y: the bisection arms the round-5 review listed, WITH the hazard put back (round-5 laziness + null-stream hipMemset):
# which of them hides it - serialised kernels, blocking launches, one chain, the layout lane pre-warmed
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06y; rm -rf $O; mkdir -p $O
H="--env YMK_DEBUG_LAZY_SPLIT=1 --env YMK_DEBUG_HAZARD_NULL_MEMSET=1"
S="python tools/stress_call.py --parallel 4 --child-timeout 120 --runs 1000 --time-budget 75"
arm() { label=$1; shift; timeout 240 $S --label $label $H "$@" --out $O/stress_$label.json < /dev/null > /dev/null; echo "$label rc $?"; }
arm hazard_again
arm hazard_serialized_kernels --env AMD_SERIALIZE_KERNEL=3
arm hazard_blocking_launches --env HIP_LAUNCH_BLOCKING=1
arm hazard_one_chain --no-concurrent
arm hazard_layout_prewarmed --prewarm
python - < /dev/null <<'PY'
import json, glob
for p in sorted(glob.glob("/root/repo/gpurun_out/r06y/stress_*.json")):
    d = json.load(open(p))
    print(d["label"], "completed", d.get("completed"), "failures", d["failures"], "cold!=warm", d.get("cold_output_differs_from_warm_output_by_stage"), "crashed", len(d["crashed"]), "proc", d.get("process_s_median"))
PY

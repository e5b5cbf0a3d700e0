# round 5, job C: the whole GPU suite in its natural order on the tree with the round-5 routes (A-stationary kernel routed,
# LayerNorm folded into its operand load, fp16 planes between reduction and 3 x 3, the cheap GELU, oracle on 32 threads),
# every slow case included (budget off), amax_check naming launches; then the A-stationary timing table and the bench with the
# routes on and off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05c; rm -rf $O; mkdir -p $O
echo "== GPU suite"
YMK_AMAX_CHECK_LEVEL=2 YMK_GPU_SLOW_BUDGET=0 YMK_HIGHWATER=$O/suite_highwater.json timeout 1500 python -m pytest tests/ -q -m gpu --durations=30 --junitxml=$O/junit.xml > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -90
echo "== astat timing"; timeout 300 python tools/astat_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/astat_timing.jsonl
echo "== bench, routes on"
timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_routes_on.json 2> $O/bench_on.err; echo "rc $?"; tail -3 $O/bench_on.err
echo "== bench, routes off (round-4 routing)"
YMK_DEBUG_OPTIONS="astat=0,parseq_no_ln_fusion=1,act_planes=0" timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_routes_off.json 2> $O/bench_off.err; echo "rc $?"; tail -3 $O/bench_off.err
python - <<'PY'
import json
for tag in ("on", "off"):
    try:
        d = json.load(open(f"gpurun_out/r05c/bench_routes_{tag}.json")); r = d["roofline"]
        print(tag, d["value"], d["unit"], "conv ms/page", r["kernel_ms_per_page"], r["bound"], r["frac"], "two-roof", r["per_launch"]["frac_of_two_roof_bound"], "dbnet", r["dbnet_conv"]["kernel_ms_per_page"], r["dbnet_conv"]["frac"])
    except Exception as e:
        print(tag, "no line:", e)
PY

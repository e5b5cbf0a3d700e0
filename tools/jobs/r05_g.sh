# round 5, job G: the fused ViT MLP - operator and recogniser tests, its timing against the launches it replaces, and the bench
# with the fusion on and off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05g; rm -rf $O; mkdir -p $O
echo "== tests"; timeout 300 python -m pytest tests/test_vit_mlp_gpu.py -q -x -s 2>&1 | grep -v "INFO\|^$" | tail -25
echo "== timing"; timeout 200 python tools/mlp_timing.py 2>&1 | grep -v amdgpu.ids | tee $O/mlp_timing.jsonl
echo "== bench, fusion on"
timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_mlp_on.json 2> $O/bench_on.err; echo "rc $?"; tail -2 $O/bench_on.err
echo "== bench, fusion off"
YMK_DEBUG_OPTIONS="parseq_no_mlp_fusion=1" timeout 600 python bench.py --gpus 1 --steps 8 --warmup 2 --no-secondary --no-cpu-baseline > $O/bench_mlp_off.json 2> $O/bench_off.err; echo "rc $?"; tail -2 $O/bench_off.err
python - <<'PY'
import json
for tag in ("on", "off"):
    try:
        d = json.load(open(f"gpurun_out/r05g/bench_mlp_{tag}.json")); r = d["roofline"]
        print(tag, d["value"], d["unit"], "conv ms/page", r["kernel_ms_per_page"], r["bound"], r["frac"], "two-roof", r["per_launch"]["frac_of_two_roof_bound"], "launches/page", r["launches_per_page"])
    except Exception as e:
        print(tag, "no line:", e)
PY

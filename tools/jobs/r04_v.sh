# round 4, job V: the per-launch table (new C-ABI entry) and the roofline pass that prices every launch against its binding roof
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04v; rm -rf $O; mkdir -p $O
timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "prof_launch_table or conv2d_matches" 2>&1 | grep -v "^$" | tail -5
timeout 300 python bench.py --roofline-only --no-cpu-baseline > $O/line_roofline_only.json 2> $O/err.log || tail -5 $O/err.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04v/line_roofline_only.json")); r=d["roofline"]
print(r["bound"], r["achieved"], r["frac"], r["mfma"]["frac"], r["hbm"]["frac_of_achievable"]); print(r["per_launch"])
PY

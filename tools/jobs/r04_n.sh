# round 4, job N: the cell detector test after its move to the same-upstream policy; whole pages against the exact kernels
# with the corrected yardstick arm; the unmodified-serve leg with a detector whose boxes survive
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04n; rm -rf $O; mkdir -p $O
echo "== cells"; timeout 600 python -m pytest tests/test_cells_gpu.py -m gpu -q 2>&1 | grep -v "INFO\|^$" | tail -4
echo "== pages eval"; SPLIT=16 CONTROL=1 timeout 900 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16_control.json 2> $O/err_pages.log || tail -8 $O/err_pages.log
cat $O/split_eval_pages_f16_control.json
SPLIT=16 ALL=1 timeout 900 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16_all.json 2> $O/err_pages2.log || tail -8 $O/err_pages2.log
cat $O/split_eval_pages_f16_all.json
echo "== unmodified serve leg"
timeout 600 python - <<'PY' 2>&1 | grep -v INFO | tail -5
import json, sys, types
sys.argv=['bench.py']
import bench, torch
from yomitoku_amd import _lib
dev=bench.rank_device(0)
sds=bench.calibrate_heads(bench.make_checkpoints('lite'), dev, bench.Page(0, dev))
pages=bench.make_pages(list(range(64)), dev)
args=types.SimpleNamespace(model_set='lite', wave=8, in_flight=4)
print(json.dumps(bench.unmodified_serve_metrics(args, dev, sds, [p.img for p in pages])))
PY

cd $GRAFT_REPO_ROOT
for r in 2 4 2 4; do
  YMK_DEC_ROWS=$r timeout 300 python bench.py --roofline-only --procs 1 --workers 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rows $r', r['achieved'], r['kernel_ms_per_page'], r['dbnet_conv']['achieved'])"
done

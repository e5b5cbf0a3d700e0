cd $GRAFT_REPO_ROOT
for r in 1 4 2 1 4; do
  YMK_DEC_ROWS=$r timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('rows $r', d['value'], d['ms_per_step'], r['achieved'], r['conv_share_of_wall'])"
done

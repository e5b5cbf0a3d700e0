# round 6, job G: the table-heavy leg against the size of a table-structure forward (16 / 32 / 64 / 128 crops); the headline
# with the greedy loop's publication from its own launch (default) and from inside the kernel (rounds 1-5)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06g; rm -rf $O; mkdir -p $O
for n in 16 32 64 128; do
  timeout 300 python tools/serve_trace.py --unmodified --steps 3 --wave 16 --in-flight 4 --max-tables $n > $O/serve_trace_unmodified_tables_$n.json 2> $O/err_$n.log || tail -3 $O/err_$n.log
  python -c "
import json; d=json.load(open('$O/serve_trace_unmodified_tables_$n.json')); print($n, d['pages_per_s'], d['table_workspace_gb'], 'GB', {k: (v['busy_frac'], v['mean_ms']) for k,v in d['stages'].items() if k in ('tables','cells','detect','layout')})"
done
for v in 1 0; do
  YMK_DEBUG_OPTIONS=ar_publish=$v timeout 400 python bench.py --gpus 1 --steps 8 --warmup 2 --no-secondary --no-roofline --no-cpu-baseline > $O/bench_ar_publish_$v.json 2> $O/bench_$v.err || tail -3 $O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_ar_publish_$v.json')); print('ar_publish=$v', d['value'], d['unit'], d['ms_per_step'])"
done

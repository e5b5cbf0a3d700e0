# round 6, job J: the batched crop pre-processing (one launch + one blob per forward, coefficient tables in array form) on the device:
# tests, then the table-heavy leg and the headline hand-overs again, and one RT-DETRv2 forward alone at 16 / 64 crops
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06j; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_imaging_gpu.py tests/test_pipeline_gpu.py tests/test_serving_gpu.py tests/test_rtdetr_gpu.py tests/test_cells_gpu.py -x -q -m gpu < /dev/null > $O/pytest_subset.log 2>&1; echo "subset rc $?"; grep -v "INFO\|^$" $O/pytest_subset.log | tail -6
for b in 16 64; do timeout 120 python tools/rtdetr_profile.py --batch $b < /dev/null 2> /dev/null | tail -1 | tee $O/rtdetr_forward_alone_b$b.txt; done
timeout 200 python tools/serve_trace.py --unmodified --fine --steps 3 --wave 16 --in-flight 4 < /dev/null > $O/serve_trace_unmodified.json 2> $O/err.log || tail -3 $O/err.log
timeout 200 python tools/serve_trace.py --steps 3 --wave 16 --in-flight 4 < /dev/null > $O/serve_trace_headline.json 2>> $O/err.log || tail -3 $O/err.log
python - < /dev/null <<'PY'
PY
python -c "
import json
for n in ('unmodified','headline'):
    d=json.load(open('$O/serve_trace_'+n+'.json')); print(n, d['units_per_page'], d['pages_per_s'], 'max tables', d['max_tables_per_forward'], d['table_workspace_gb'], 'GB')
    for k,v in sorted(d['stages'].items(), key=lambda kv:-kv[1]['busy_frac'])[:7]: print('   ', k, v['busy_frac'], v['mean_ms'])
    for k,v in sorted(d.get('fine_ms',{}).items(), key=lambda kv:-kv[1]['total_per_wave'])[:6]: print('    fine', k, v['total_per_wave'])
" < /dev/null

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03f; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_serving_gpu.py tests/test_pipeline_gpu.py -m gpu -q -x 2>&1 | grep -v INFO | tail -6
for f in 3 4 6; do
  timeout 300 python tools/serve_trace.py --steps 3 --in-flight $f --fine > $O/fine_f$f.json 2>> $O/err.log || tail -5 $O/err.log
done
timeout 300 python tools/serve_trace.py --steps 4 --in-flight 4 > $O/plain_f4.json 2>> $O/err.log
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/serve_trace.py --steps 2 --in-flight 4 > $O/rocprof_host.json 2> $O/rocprof.err || tail -5 $O/rocprof.err
python tools/gpu_timeline.py $O/kt $O/gpu_timeline.json > /dev/null 2>> $O/rocprof.err || tail -5 $O/rocprof.err
rm -rf $O/kt
for f in $O/*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f'))
if 'stages' in d:
    print(d['pages_per_s'], {k:(v['busy_frac'],v['mean_ms']) for k,v in d['stages'].items()})
    for k,v in (d.get('fine_ms') or {}).items(): print('     ',k,v)
else:
    print({k:v for k,v in d.items() if k not in ('largest_gaps','kernel_time_ms_by_family')}); print(d['largest_gaps'][:6])"; done

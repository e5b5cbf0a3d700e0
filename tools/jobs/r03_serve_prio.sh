cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03e; rm -rf $O; mkdir -p $O
timeout 300 python tools/serve_trace.py --steps 3 > $O/base_f3.json 2>> $O/err.log || tail -5 $O/err.log
timeout 300 python tools/serve_trace.py --steps 3 --in-flight 4 > $O/base_f4.json 2>> $O/err.log
YMK_STAGE_PRIORITY=recognize:-1 timeout 300 python tools/serve_trace.py --steps 3 --in-flight 4 > $O/prio_rec_f4.json 2>> $O/err.log
YMK_STAGE_PRIORITY=recognize:-1,layout:-1 timeout 300 python tools/serve_trace.py --steps 3 --in-flight 4 > $O/prio_rec_lay_f4.json 2>> $O/err.log
YMK_STAGE_PRIORITY=detect:-1 timeout 300 python tools/serve_trace.py --steps 3 --in-flight 4 > $O/prio_det_f4.json 2>> $O/err.log
timeout 300 python tools/serve_trace.py --steps 3 --in-flight 6 > $O/base_f6.json 2>> $O/err.log
timeout 300 python tools/serve_trace.py --steps 3 --in-flight 4 --wave 16 > $O/base_w16f4.json 2>> $O/err.log
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/serve_trace.py --steps 2 --in-flight 4 > $O/rocprof_host.json 2> $O/rocprof.err || tail -5 $O/rocprof.err
python tools/gpu_timeline.py $O/kt $O/gpu_timeline.json > /dev/null 2>> $O/rocprof.err || tail -5 $O/rocprof.err
rm -rf $O/kt
for f in $O/*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f'))
if 'stages' in d:
    print(d['pages_per_s'], {k:(v['busy_frac'],v['mean_ms']) for k,v in d['stages'].items()}); print('   gc', d['gc']['collections']['2'])
else:
    print({k:v for k,v in d.items() if k not in ('largest_gaps','kernel_time_ms_by_family')}); print(d['largest_gaps'][:4])"; done

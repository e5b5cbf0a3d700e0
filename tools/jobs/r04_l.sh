# round 4, job L: where a serve() job spends its time on the fp16 path: host stage trace, device kernel timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04l; rm -rf $O; mkdir -p $O
timeout 300 python tools/serve_trace.py --steps 4 --in-flight 4 --fine > $O/host_trace.json 2> $O/host_trace.err || tail -5 $O/host_trace.err
python -c "
import json
d=json.load(open('$O/host_trace.json'))
print(d['pages_per_s'], {k:(v['busy_frac'],v['mean_ms']) for k,v in d['stages'].items()})
for k,v in (d.get('fine_ms') or {}).items(): print('   ',k,v)
"
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/serve_trace.py --steps 2 --in-flight 4 > $O/host_trace_rocprof.json 2> $O/rocprof.err || tail -5 $O/rocprof.err
python tools/gpu_timeline.py $O/kt $O/gpu_timeline.json 2>> $O/rocprof.err | cut -c1-1500 || tail -5 $O/rocprof.err
rm -rf $O/kt

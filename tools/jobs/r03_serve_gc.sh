cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03d; rm -rf $O; mkdir -p $O
for mode in default off freeze; do
  timeout 300 python tools/serve_trace.py --steps 3 --gc $mode > $O/gc_$mode.json 2>> $O/err.log || tail -5 $O/err.log
done
timeout 300 python tools/serve_trace.py --steps 3 --gc off --in-flight 4 > $O/gc_off_f4.json 2>> $O/err.log
timeout 300 python tools/serve_trace.py --steps 3 --gc off --in-flight 2 > $O/gc_off_f2.json 2>> $O/err.log
for f in $O/*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f'))
print(d['pages_per_s'], {k:(v['busy_frac'],v['mean_ms']) for k,v in d['stages'].items()})
print('   gc', d['gc'])"; done

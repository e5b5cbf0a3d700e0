cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03c; rm -rf $O; mkdir -p $O
timeout 300 python tools/serve_trace.py --steps 3 --fine --sample > $O/fine.json 2> $O/err.log || tail -5 $O/err.log

for f in $O/*.json; do echo "== $f"; python -c "
import json,sys
d=json.load(open('$f'))
print(d['pages_per_s'], {k:(v['busy_frac'],v['mean_ms']) for k,v in d['stages'].items()})
for k,v in (d.get('fine_ms') or {}).items(): print('   ',k,v)
s=d.get('sampler')
if s:
    print('gaps', s['sampler_gaps_over_4ms'], s['sampler_gap_total_ms'])
    for g in s['largest_sampler_gaps']: print('   gap', g)
    for t,fr in s['top_frames_per_thread'].items(): print('  ',t, fr)
"; done

# round 6, job Z: detector forwards of 16 pages instead of 8 (A/B, consecutive short runs of the bench)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06z; rm -rf $O; mkdir -p $O
run() { n=$1; tag=$2; timeout 300 python -c "
import sys
import yomitoku_amd.text_detector as t
t.TextDetector.MAX_PAGES_PER_FORWARD = $n
sys.argv = ['bench.py', '--gpus', '1', '--steps', '8', '--warmup', '2', '--no-secondary', '--no-roofline', '--no-cpu-baseline']
import bench
bench.main()
" < /dev/null > $O/bench_det_$n_$tag.json 2> $O/err_$n_$tag.log; python -c "
import json; d=json.load(open('$O/bench_det_$n_$tag.json')); print('detector forwards of $n pages ($tag):', d['value'], d['unit'], d['highwater'].get('vram_peak_gb'))" < /dev/null; }
run 8 a; run 16 a; run 8 b; run 16 b

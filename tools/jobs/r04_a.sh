# round 4, first GPU job: fp16-split convolutions (numerics + sweep + whole pages), the persistent fp32 tile loop (first run),
# kernel-time breakdown of the default model set
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
echo "== split tests"; timeout 300 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -x -s 2>&1 | grep -v "INFO\|^$" | tail -16
echo "== persistent tests"; YMK_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "persistent" 2>&1 | grep -v "INFO\|^$" | tail -4
echo "== sweep f16"; VARIANTS="0,b2t3,b16t3,b16t1,b16t2,b16t4,b16t11" REPS=5 timeout 400 python tools/conv_sweep.py > $O/sweep_f16.txt 2> $O/err0.log || tail -5 $O/err0.log
cut -c1-260 $O/sweep_f16.txt
echo "== sweep persistent"; ONLY="64->256|128->512|256->1024|512->2048|parseq|dec 1x1|l1 1x1|l1 3x3|dec 3x3" VARIANTS="0/27,0/59,0/27,0/59" REPS=7 timeout 150 python tools/conv_sweep.py > $O/sweep_persistent.txt 2> $O/err1.log || tail -5 $O/err1.log
cut -c1-200 $O/sweep_persistent.txt
echo "== pages eval (det+layout+table f16)"; SPLIT=16 timeout 600 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16.json 2> $O/err2.log || tail -8 $O/err2.log
cat $O/split_eval_pages_f16.json
echo "== pages eval (all four f16)"; SPLIT=16 ALL=1 timeout 600 python tools/split_eval_pages.py 48 > $O/split_eval_pages_f16_all.json 2> $O/err3.log || tail -8 $O/err3.log
cat $O/split_eval_pages_f16_all.json
echo "== parseq eval"; timeout 600 python tools/split_eval_parseq.py > $O/split_eval_parseq.json 2> $O/err4.log || tail -8 $O/err4.log
cat $O/split_eval_parseq.json
echo "== default model set under rocprofv3"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/prof_default -o default -- python bench.py --model-set default --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_default_set.json 2> $O/err5.log || tail -5 $O/err5.log
cat $O/bench_default_set.json | cut -c1-600
f=$(ls $O/prof_default/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $O/default_set_kernel_stats.csv && head -25 $f | cut -c1-200
rm -rf $O/prof_default

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03i; rm -rf $O; mkdir -p $O
timeout 300 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -x 2>&1 | tail -2
ONLY="l4 3x3|dec 3x3|l4 1x1 1024|l2 3x3|l1 1x1 64->256|dec 1x1|rtdetr enc|parseq fc1|parseq fc2" VARIANTS="0,b2t3,b2t5,b2t6,b2t7,b2t8,b2t9,b3t2,b3t6,b3t8,b3t9" REPS=5 timeout 900 python tools/conv_sweep.py > $O/sweep_split2.txt 2> $O/err.log || tail -5 $O/err.log
cat $O/sweep_split2.txt

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03m; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_serving_gpu.py tests/test_conv_split_gpu.py -m gpu -q -x 2>&1 | grep -v "INFO\|^$" | tail -4
ONLY="l4 3x3|l4 1x1|l3 3x3|dec 1x1|rtdetr enc|parseq fc1|parseq head 192" VARIANTS="0,b2t3,b2t11,b2t2" REPS=5 timeout 600 python tools/conv_sweep.py > $O/sweep4.txt 2> $O/err.log || tail -5 $O/err.log
cat $O/sweep4.txt

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03g; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -x -s 2>&1 | grep -v INFO | tail -14
VARIANTS="0,b3,b3t3,b3t1,b3t2,b2,b2t3,b2t2" REPS=5 timeout 900 python tools/conv_sweep.py > $O/sweep_split.txt 2> $O/err.log || tail -5 $O/err.log
cat $O/sweep_split.txt

cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03k; rm -rf $O; mkdir -p $O
ONLY="l4 3x3|l4 1x1 1024|dec 1x1|parseq fc2" VARIANTS="0,b2t3,b2t10,b3t2,b3t10" REPS=5 timeout 600 python tools/conv_sweep.py > $O/sweep3.txt 2> $O/err.log || tail -5 $O/err.log
cat $O/sweep3.txt
timeout 1500 python -m pytest tests/test_baseline_configs_gpu.py tests/test_rtdetr_gpu.py tests/test_cells_gpu.py -m gpu -q -x -s 2>&1 | grep -v "INFO\|^$" | tail -25

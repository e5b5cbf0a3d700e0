# round 4, job AD: edge shapes of the candidate kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04ad; rm -rf $O; mkdir -p $O
timeout 30 python -m pytest tests/test_conv_astat_gpu.py -m gpu -q -k "edge or refuses" 2>&1 | grep -v "^$" | tail -30 > $O/test.log; tail -25 $O/test.log

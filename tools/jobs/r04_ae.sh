# round 4, job AE: the candidate kernel's tests once more on the final library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 14 python -m pytest tests/test_conv_astat_gpu.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -4

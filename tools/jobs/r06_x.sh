# round 6, job X: the tests that touch the last host-side change (the page dealer's failure paths), on the device
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_distributed_gpu.py tests/test_serving_gpu.py -x -q -m gpu < /dev/null 2>&1 | grep -v INFO | tail -4

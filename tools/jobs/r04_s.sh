# round 4, job S: the RCCL leg on the one GPU there is (process group of one rank, backend nccl)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04s; rm -rf $O; mkdir -p $O
timeout 400 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q 2>&1 | grep -v "^$" | tail -15 | tee $O/test.log

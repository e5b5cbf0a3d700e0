# round 6, job E: e2e vs the free-running oracle on table-rich checkpoints (six tables asked of the calibration page), then the GPU test of the same
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06e; rm -rf $O; mkdir -p $O
timeout 900 python tools/e2e_oracle_eval.py --pages 16 --first-seed 40 --layout-targets 6,1,6,2,1,1 --out $O/e2e_oracle_eval_16_pages_table_rich.json 2> $O/e2e.err; echo "rc $?"; grep -v INFO $O/e2e.err | tail -20
timeout 600 python -m pytest tests/test_e2e_oracle_gpu.py -q -m gpu -x -s 2>&1 | grep -v INFO | tail -8

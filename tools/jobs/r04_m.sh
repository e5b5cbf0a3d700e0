# round 4, job M: the test files the final run did not reach (it stopped at its first failure), with the failure's traceback
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04m; rm -rf $O; mkdir -p $O
echo "== cells"; timeout 600 python -m pytest tests/test_cells_gpu.py -m gpu -q -x 2>&1 | grep -v "INFO" | grep -v "^$" | tail -40
echo "== cells, exact fp32"; YMK_CONV_SPLIT=0 timeout 600 python -m pytest tests/test_cells_gpu.py -m gpu -q -x 2>&1 | grep -v "INFO" | grep -v "^$" | tail -5
echo "== rest"; timeout 1200 python -m pytest tests/test_imaging_gpu.py tests/test_ops_gpu.py tests/test_pipeline_gpu.py tests/test_seq_ops_gpu.py -m gpu -q 2>&1 | grep -v "INFO\|^$" | tail -15

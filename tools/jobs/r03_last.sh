cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_baseline_configs_gpu.py::test_whole_page_schema_vs_oracle_chain tests/test_serving_gpu.py -m gpu -q -x -s 2>&1 | grep -v "INFO\|^$" | tail -8

# round 4, last job: smoke() and the page-level parity tests on the final library
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 75 python -m pytest tests/test_pipeline_gpu.py tests/test_abi.py -x -q 2>&1 | grep -v "^$" | tail -4

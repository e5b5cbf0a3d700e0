# round 5, job Q: one whole-page test repeated in fresh processes (a failure after an AR-loop change that was then taken back:
# how often does the tree WITHOUT that change fail it?); the element lists of both sides are printed on failure
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05q; rm -rf $O; mkdir -p $O
for r in ${RUNS:-1 2 3 4 5 6}; do
timeout 60 python -m pytest "tests/test_baseline_configs_gpu.py::test_whole_page_schema_vs_oracle_chain[page_hw1]" -x -q -m gpu > $O/run_$r.log 2>&1; rc=$?; echo "run $r rc $rc"
if [ $rc != 0 ]; then grep -E "AssertionError|elapsed_time" $O/run_$r.log | cut -c1-300 | head -8; fi
done

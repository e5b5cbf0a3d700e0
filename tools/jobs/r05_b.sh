# round 5, job B: which launches carry a max|x| record below the truth (amax_check level 2), the CPU oracle's thread scaling on
# the box, and the part of the GPU suite job A never reached (it stopped at its first failure)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05b; rm -rf $O; mkdir -p $O
echo "== amax diagnosis (serve)"; timeout 300 python tools/diag/amax_below.py serve 2>&1 | grep -v "INFO\|amdgpu.ids" | sort | uniq -c | sort -rn | head -40
echo "== amax diagnosis (nets alone)"; timeout 300 python tools/diag/amax_below.py nets 2>&1 | grep -v "INFO\|amdgpu.ids" | sort | uniq -c | sort -rn | head -30
echo "== oracle threads"; timeout 600 python tools/diag/oracle_threads.py 2>&1 | grep -v amdgpu.ids
echo "== rest of the suite"
YMK_HIGHWATER=$O/suite_highwater_rest.json timeout 1200 python -m pytest tests/ -q -m gpu --durations=15 --junitxml=$O/junit_rest.xml \
  --ignore=tests/test_baseline_configs_gpu.py --ignore=tests/test_cells_gpu.py --ignore=tests/test_conv_astat_gpu.py > $O/pytest_rest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest_rest.log | tail -60

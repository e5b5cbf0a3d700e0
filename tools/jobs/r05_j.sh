# round 5, job J: the whole GPU suite on the final tree (the routes test counts the fused MLP's launches now)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05j; rm -rf $O; mkdir -p $O
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=8 --junitxml=$O/junit.xml > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -16

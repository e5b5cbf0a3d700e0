# round 6, job AG: the whole GPU suite on the final tree (job AF stopped at a wrong assertion of the new test itself)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06ag; rm -rf $O; mkdir -p $O
YMK_HIGHWATER=$O/suite_highwater.json timeout 1200 python -m pytest tests/ -x -q -m gpu --durations=8 --junitxml=$O/junit.xml < /dev/null > $O/pytest.log 2>&1
echo "suite rc $?"; grep -v "INFO\|^$" $O/pytest.log | tail -14

# round 5, job O: the AR loop's open-rows word published from a launch of its own - the recogniser's tests, then the greedy kernel's
# time in a kernel trace of the serial pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05o; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_parseq_gpu.py tests/test_pipeline_gpu.py tests/test_serving_gpu.py -x -q -m gpu > $O/pytest.log 2>&1; echo "tests rc $?"; tail -3 $O/pytest.log
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python bench.py --roofline-only --no-cpu-baseline > $O/line.json 2> $O/kt.log || tail -5 $O/kt.log
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \; ; rm -rf $O/kt
grep -i "greedy\|publish" $O/kernel_stats.csv | sed 's/(.*)"//' | cut -c1-120

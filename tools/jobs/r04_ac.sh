# round 4, job AC: candidate kernel with the residual values of 16 accumulator rows fetched together (K <= 128): parity again, timing
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04ac; rm -rf $O; mkdir -p $O
timeout 40 python -m pytest tests/test_conv_astat_gpu.py -m gpu -q 2>&1 | grep -v "^$" | tail -8 > $O/test.log; tail -3 $O/test.log
ONLY="l1 64|l2 128" timeout 25 python tools/astat_timing.py > $O/astat_timing.jsonl 2> $O/timing.err; cat $O/astat_timing.jsonl

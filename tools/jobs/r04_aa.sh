# round 4, job AA: the candidate A-stationary kernel - parity against fp64 and bit-identity against the library's fp16 kernels,
# then its time on three short-K shapes of the analyzer next to the library's
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04aa; rm -rf $O; mkdir -p $O
timeout 55 python -m pytest tests/test_conv_astat_gpu.py -m gpu -q 2>&1 | grep -v "^$" | tail -25 > $O/test.log; tail -12 $O/test.log
ONLY="fc1|qkv|l1 64" timeout 30 python tools/astat_timing.py > $O/astat_timing.jsonl 2> $O/timing.err; cat $O/astat_timing.jsonl; tail -3 $O/timing.err

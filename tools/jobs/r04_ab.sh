# round 4, job AB: the candidate A-stationary kernel on the other three short-K shapes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04ab; rm -rf $O; mkdir -p $O
ONLY="proj|l2 128|dec 256" timeout 40 python tools/astat_timing.py > $O/astat_timing.jsonl 2> $O/timing.err; cat $O/astat_timing.jsonl; tail -2 $O/timing.err

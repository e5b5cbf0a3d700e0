cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03n; rm -rf $O; mkdir -p $O
timeout 900 python tools/split_eval_parseq.py > $O/split_eval_parseq.json 2> $O/err.log || tail -8 $O/err.log
cat $O/split_eval_parseq.json

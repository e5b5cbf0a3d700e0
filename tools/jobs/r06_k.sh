# round 6, job K: the table-structure stage of a wave alone, part by part
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06k; rm -rf $O; mkdir -p $O
timeout 200 python tools/tables_stage_timing.py < /dev/null > $O/tables_stage_timing.json 2> $O/err.log || tail -5 $O/err.log
cat $O/tables_stage_timing.json

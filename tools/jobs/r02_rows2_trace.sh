cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02g; rm -rf $O; mkdir -p $O
YMK_DEC_ROWS=2 timeout 400 rocprofv3 --kernel-trace -d $O/kt -o kt -- python bench.py --roofline-only --procs 1 --workers 1 --no-cpu-baseline > $O/line.json 2> $O/kt.log || tail -5 $O/kt.log
ls -la $O/kt

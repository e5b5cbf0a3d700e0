cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_parseq_gpu.py -x -q 2>&1 | tail -2
O=gpurun_out/r02i; rm -rf $O; mkdir -p $O
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/mfma -o m -- python bench.py --roofline-only --procs 1 --workers 1 --no-cpu-baseline > $O/line.json 2> $O/m.log || tail -3 $O/m.log
ls -la $O/mfma

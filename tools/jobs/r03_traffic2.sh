# HBM traffic of the conv launches after the tile change: FETCH_SIZE / WRITE_SIZE passes of the serial roofline pass
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03t; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 95 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 95 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_f.json $O/fetch $O/write $O/traffic.json
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete; rm -rf $O/fetch $O/write

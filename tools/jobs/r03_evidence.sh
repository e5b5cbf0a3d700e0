# round-3 evidence of the fp32 headline path: kernel trace of the serial roofline pass (cross-check of the live HIP events),
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and MFMA-busy per kernel instantiation
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03p; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
timeout 500 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 500 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/mfma -o m -- $B > $O/line_m.json 2> $O/m.log || tail -5 $O/m.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json $O/fetch $O/write $O/traffic.json
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
python tools/pmc_aggregate.py sum $O/mfma $O/mfma_by_kernel.csv
find $O -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
rm -rf $O/kt $O/fetch $O/write $O/mfma
du -sh $O; head -12 $O/kernel_stats.csv

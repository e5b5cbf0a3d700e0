# round 4, job Y: kernel trace + MFMA-busy passes of the roofline pass as it runs now (waves of 16 pages), for the cross-check of
# the live HIP events, the HBM GB/s table (with job X's FETCH / WRITE passes) and the MFMA-busy table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04y; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json | cut -c1-700
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -o m -- $B > $O/line_m.json 2> $O/m.log || tail -5 $O/m.log
python tools/pmc_aggregate.py sum $O/mfma $O/mfma_by_kernel.csv
rm -rf $O/kt $O/mfma

# round 4, job G: parity tests of the nets on the widened split dispatch, then the evidence passes of the headline path:
# kernel trace of the serial roofline pass (cross-check of the live HIP events), HBM traffic (FETCH_SIZE / WRITE_SIZE, separate
# passes), MFMA-busy per kernel instantiation
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04g; rm -rf $O; mkdir -p $O
echo "== tests"; timeout 1200 python -m pytest tests/test_conv_split_gpu.py tests/test_parseq_gpu.py tests/test_rtdetr_gpu.py tests/test_dbnet_gpu.py tests/test_serving_gpu.py -m gpu -q 2>&1 | grep -v "INFO\|^$" | tail -12
B="python bench.py --roofline-only --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_kt.json $O/fetch $O/write $O/traffic.json
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -o m -- $B > $O/line_m.json 2> $O/m.log || tail -5 $O/m.log
python tools/pmc_aggregate.py sum $O/mfma $O/mfma_by_kernel.csv
find $O -name "*_kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*.db" -delete
rm -rf $O/kt $O/fetch $O/write $O/mfma
du -sh $O; head -14 $O/kernel_stats.csv | cut -c1-180; head -12 $O/mfma_by_kernel.csv | cut -c1-200

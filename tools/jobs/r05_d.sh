# round 5, job D: evidence for the round-5 routes - rocprofv3 kernel trace of the serial roofline pass (cross-check of the live
# events), MFMA-busy and instruction-mix PMC passes with the fp16 planes on and off, HBM traffic (FETCH / WRITE passes), the
# per-layer two-roof table on / off, and the host tests touched since job C
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05d; rm -rf $O; mkdir -p $O
B="python bench.py --roofline-only --no-cpu-baseline"
echo "== touched tests"; timeout 300 python -m pytest tests/test_split_robustness_gpu.py tests/test_serving_gpu.py tests/test_routes_gpu.py -q -x 2>&1 | grep -v "INFO\|^$" | tail -5
echo "== kernel trace"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- $B > $O/line_kt.json 2> $O/kt.log || tail -5 $O/kt.log
python tools/roofline_crosscheck.py $O/line_kt.json $O/kt $O/crosscheck.json | cut -c1-700
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
echo "== MFMA busy"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/mfma -o m -- $B > $O/line_m.json 2> $O/m.log || tail -5 $O/m.log
python tools/pmc_aggregate.py sum $O/mfma $O/mfma_by_kernel.csv
echo "== instruction mix, planes on / off"
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/mix_on -o m -- $B > /dev/null 2> $O/mix_on.log || tail -5 $O/mix_on.log
python tools/pmc_aggregate.py sum $O/mix_on $O/mix_planes_on_by_kernel.csv
YMK_DEBUG_OPTIONS="act_planes=0" timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/mix_off -o m -- $B > /dev/null 2> $O/mix_off.log || tail -5 $O/mix_off.log
python tools/pmc_aggregate.py sum $O/mix_off $O/mix_planes_off_by_kernel.csv
echo "== traffic"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -o f -- $B > $O/line_f.json 2> $O/f.log || tail -5 $O/f.log
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -o w -- $B > $O/line_w.json 2> $O/w.log || tail -5 $O/w.log
python tools/roofline_crosscheck.py --traffic-only $O/line_f.json $O/fetch $O/write $O/traffic.json | cut -c1-600
python tools/pmc_aggregate.py sum $O/fetch $O/fetch_by_kernel.csv; python tools/pmc_aggregate.py sum $O/write $O/write_by_kernel.csv
rm -rf $O/kt $O/mfma $O/mix_on $O/mix_off $O/fetch $O/write
echo "== two-roof table by layer, routes on / planes off / all off"
YMK_DEBUG_OPTIONS="prof_dump=1" timeout 120 $B > /dev/null 2> $O/dump_on.txt; python tools/two_roof.py $O/dump_on.txt $O/two_roof_on.md 3 | tail -2
YMK_DEBUG_OPTIONS="prof_dump=1,act_planes=0" timeout 120 $B > /dev/null 2> $O/dump_planes_off.txt; python tools/two_roof.py $O/dump_planes_off.txt $O/two_roof_planes_off.md 3 | tail -2
YMK_DEBUG_OPTIONS="prof_dump=1,act_planes=0,astat=0,parseq_no_ln_fusion=1" timeout 120 $B > /dev/null 2> $O/dump_all_off.txt; python tools/two_roof.py $O/dump_all_off.txt $O/two_roof_all_off.md 3 | tail -2
gzip -f $O/dump_on.txt $O/dump_planes_off.txt $O/dump_all_off.txt
ls -la $O

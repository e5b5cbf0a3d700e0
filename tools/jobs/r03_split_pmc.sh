cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03j; rm -rf $O; mkdir -p $O
export ONLY="l4 3x3|dec 1x1" VARIANTS="0,b2t3,b3t2" REPS=4
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace -d $O/p1 -o p -- python tools/conv_sweep.py > $O/p1.txt 2> $O/p1.err || tail -3 $O/p1.err
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace -d $O/p2 -o p -- python tools/conv_sweep.py > $O/p2.txt 2> $O/p2.err || tail -3 $O/p2.err
python tools/pmc_aggregate.py sum $O/p1 $O/pass1.csv; python tools/pmc_aggregate.py sum $O/p2 $O/pass2.csv
rm -rf $O/p1 $O/p2
grep -E "conv_igemm" $O/pass1.csv | head -40; grep -E "conv_igemm" $O/pass2.csv | head -40

# round 4, job U: where the cycles of the fp16 kernels go on the HBM-bound 1 x 1 layers (the shapes that hold the analyzer's
# conv roofline down): wave occupancy / waiting / issue counters, instruction mix, L2 hit rate - separate PMC passes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04u; rm -rf $O; mkdir -p $O
export ONLY="l1 1x1 256->64|dec 1x1 256->256|l1 1x1 64->256|l2 1x1 128->512|l4 3x3" VARIANTS="b16t0" REPS=4
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/p1 -o p -- python tools/conv_sweep.py > $O/p1.txt 2> $O/p1.err || tail -3 $O/p1.err
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU --kernel-trace --output-format csv -d $O/p2 -o p -- python tools/conv_sweep.py > $O/p2.txt 2> $O/p2.err || tail -3 $O/p2.err
timeout 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/p3 -o p -- python tools/conv_sweep.py > $O/p3.txt 2> $O/p3.err || tail -3 $O/p3.err
for k in 1 2 3; do python tools/pmc_aggregate.py bygrid $O/p$k $O/pass$k.csv; done
rm -rf $O/p1 $O/p2 $O/p3
cat $O/p1.txt | grep -v INFO | tail -7
grep -c . $O/pass1.csv $O/pass2.csv $O/pass3.csv

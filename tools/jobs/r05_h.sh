# round 5, job H: where the cycles of the fused ViT MLP kernel go - two PMC passes over tools/mlp_timing.py
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05h; rm -rf $O; mkdir -p $O
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -o p -- python tools/mlp_timing.py > $O/p1.txt 2> $O/p1.err || tail -3 $O/p1.err
timeout 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $O/p2 -o p -- python tools/mlp_timing.py > $O/p2.txt 2> $O/p2.err || tail -3 $O/p2.err
for k in 1 2; do python tools/pmc_aggregate.py bygrid $O/p$k $O/pass$k.csv; done
rm -rf $O/p1 $O/p2
grep "k_vit_mlp" $O/pass1.csv | cut -c1-300; grep "k_vit_mlp" $O/pass2.csv | cut -c1-300

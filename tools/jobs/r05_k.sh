# round 5, job K: the standalone microbenchmarks (tools/diag/*.hip, built into scratch/ with the command in each file's header)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05k; mkdir -p $O
timeout 120 scratch/mfma_valu_overlap > $O/mfma_valu_overlap.jsonl 2> $O/err.txt; echo "rc $?"
timeout 120 scratch/mfma_valu_overlap second > $O/mfma_valu_overlap_set2.jsonl 2>> $O/err.txt; echo "rc $?"
timeout 120 scratch/dma_mfma_overlap > $O/dma_mfma_overlap.jsonl 2>> $O/err.txt; echo "rc $?"
tail -3 $O/err.txt

# round 5, job L: the LDS-DMA fp16-split kernels (128-row form = conv_split_tile 21, the experiment's 256 x 256 form = 22) against
# copies of themselves with one part of the loop removed or replaced (-DYMK_ABLATE=n).  The libraries come from
# `bash tools/diag/conv_dma_wide/build.sh 1 2 3 4 5 6 7 8` (scratch/libymk_wide.so, scratch/libymk_ablate_n.so: the product
# library with the experiment's kernel file - nothing of this is in the product build).  Summary: profiles/r05_conv_dma_ablation.md
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05l; mkdir -p $O
export ONLY="${ONLY:-3x3 512|l3 3x3|2048->512|parseq-large fc2}" VARIANTS=${VARIANTS:-b16t21,b16t22} REPS=5
YMK_LIB=scratch/libymk_wide.so timeout 300 python tools/conv_sweep.py > $O/full.txt 2> $O/err_full.txt; echo "full rc $?"
for n in ${ABL:-1 2 3 4 5 6 7 8}; do
  [ -f scratch/libymk_ablate_$n.so ] || continue
  YMK_LIB=scratch/libymk_ablate_$n.so timeout 200 python tools/conv_sweep.py > $O/ablate_$n.txt 2> $O/err_$n.txt; echo "ablate $n rc $?"
done
for f in $O/full.txt $O/ablate_*.txt; do echo "== $f"; cut -c1-150 $f; done

# round 5, job M: A/B of another build of the library (ALT, default the experiment's wide form) against the product library:
# same shapes, alternating processes.  VARIANTS as tools/conv_sweep.py (b16 = fp16 planes, automatic tile; t21 / t22 / t3 = the
# 128-row LDS-DMA form / the 256 x 256 form (ALT only) / the register-staged kernel)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05m; rm -rf $O; mkdir -p $O
export ONLY="${ONLY:-3x3|256->64|2048->512|1024->256}" VARIANTS=${VARIANTS:-b16t21} REPS=7
for r in 1 2; do
timeout 300 python tools/conv_sweep.py > $O/base_$r.txt 2> $O/err.txt; echo "base rc $?"
YMK_LIB=${ALT:-scratch/libymk_wide.so} timeout 300 python tools/conv_sweep.py > $O/alt_$r.txt 2>> $O/err.txt; echo "alt rc $?"
done
for f in base_1 alt_1 base_2 alt_2; do echo "== $f"; cat $O/$f.txt | cut -c1-110; done

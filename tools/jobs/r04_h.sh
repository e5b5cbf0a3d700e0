# round 4, job H: the two-blocks-per-CU forms of the LDS-DMA kernel (conv_split_tile 21 / 22): bit-identity, sweep
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04h; rm -rf $O; mkdir -p $O
echo "== dma tests"; timeout 600 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -k "lds_dma" 2>&1 | grep -v "INFO\|^$" | tail -6
echo "== sweep"; VARIANTS="b16t3,b16t20,b16t21,b16t22" REPS=5 timeout 400 python tools/conv_sweep.py > $O/sweep_dma2.txt 2> $O/err0.log || tail -5 $O/err0.log
cut -c1-200 $O/sweep_dma2.txt

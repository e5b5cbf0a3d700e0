# round 4, job E: first run of the LDS-DMA form of the fp16-split kernel (conv_split_tile 20): bit-identity with the
# register-staged kernel, then the sweep against it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04e; rm -rf $O; mkdir -p $O
echo "== dma tests"; timeout 500 python -m pytest tests/test_conv_split_gpu.py -m gpu -q -s -k "lds_dma" 2>&1 | grep -v "INFO\|^$" | tail -30
echo "== sweep"; VARIANTS="b16t3,b16t20,b16t3,b16t20" REPS=5 timeout 300 python tools/conv_sweep.py > $O/sweep_dma.txt 2> $O/err0.log || tail -5 $O/err0.log
cut -c1-200 $O/sweep_dma.txt

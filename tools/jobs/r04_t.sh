# round 4, job T: the encoder GEMMs of the default recogniser (D = 768) on the register-staged fp16 kernel and the LDS-DMA forms
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04t; rm -rf $O; mkdir -p $O
ONLY="parseq-large" VARIANTS="0,b16t0,b16t3,b16t20,b16t21" timeout 200 python tools/conv_sweep.py 2>&1 | grep -v INFO | tee $O/sweep_parseq_large.txt

# second pass: swizzled 128 x 64 tiles (default now) against the round-2 kernels, 64-wide tiles forced on the wide shapes, model-level A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03f; rm -rf $O; mkdir -p $O
VARIANTS="0/3,0/27,8/27,0/11,8/11,0/3" REPS=7 timeout 150 python tools/conv_sweep.py > $O/sweep_swizzle.txt 2> $O/err1.log || tail -5 $O/err1.log
cat $O/sweep_swizzle.txt
timeout 150 python tools/roofline_ab.py "conv_fast=3" "conv_fast=27" "conv_fast=27,conv_variant=8" "conv_fast=3" "conv_fast=27" --dump $O/launches > $O/roofline_ab.txt 2> $O/err3.log || tail -5 $O/err3.log
cat $O/roofline_ab.txt
timeout 150 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "epilogue" 2>&1 | grep -v "INFO\|^$" | tail -4

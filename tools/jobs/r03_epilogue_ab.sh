# direct epilogue / swizzled K tiles / staggered start: kernel sweep, bit-identity tests, model-level A/B of the serial conv pass
# (record of the job behind profiles/r03_conv_sweep_epilogue_direct_swizzle.txt and _staggered_start.txt: the conv_stagger_* options it
# sweeps were removed from the library after it showed no effect, and `conv_fast` 3 was the default at the time)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03e; rm -rf $O; mkdir -p $O
VARIANTS="0,0/7,0/11,0/15,2/7,4/7,8/7,8/15" REPS=5 timeout 150 python tools/conv_sweep.py > $O/sweep_epilogue.txt 2> $O/err1.log || tail -5 $O/err1.log
cat $O/sweep_epilogue.txt
ONLY="64->256|128->512|256->1024|parseq|dec 1x1|l1 1x1|l1 3x3" VARIANTS="0/7,0/7/conv_stagger_bit=0,0/7/conv_stagger_bit=2,0/7/conv_stagger_bit=4,0/7/conv_stagger_bit=5,0/7/conv_stagger_bit=5;conv_stagger_pct=25,0/15/conv_stagger_bit=5" REPS=5 timeout 90 python tools/conv_sweep.py > $O/sweep_stagger.txt 2> $O/err2.log || tail -5 $O/err2.log
cat $O/sweep_stagger.txt
timeout 150 python tools/roofline_ab.py "conv_fast=3" "conv_fast=7" "conv_fast=15" "conv_fast=15,conv_variant=8" "conv_fast=3" --dump $O/launches > $O/roofline_ab.txt 2> $O/err3.log || tail -5 $O/err3.log
cat $O/roofline_ab.txt
timeout 150 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "epilogue or direct or swizzled" 2>&1 | grep -v "INFO\|^$" | tail -4
YMK_DEBUG_OPTIONS="conv_fast=15" timeout 120 python -m pytest tests/test_dbnet_gpu.py tests/test_rtdetr_gpu.py tests/test_parseq_gpu.py -m gpu -q -x -k "golden or batch_consistency" 2>&1 | grep -v "INFO\|^$" | tail -4

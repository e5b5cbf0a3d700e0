# first run of conv_igemm_persist (conv_fast bit 5): bit-identity, then the sweep and the model-level A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04a; rm -rf $O; mkdir -p $O
YMK_EXPERIMENTAL=1 timeout 200 python -m pytest tests/test_ops_gpu.py -m gpu -q -x -k "persistent" 2>&1 | grep -v "INFO\|^$" | tail -4
ONLY="64->256|128->512|256->1024|512->2048|parseq|dec 1x1|l1 1x1|l1 3x3|dec 3x3" VARIANTS="0/27,0/59,0/27,0/59" REPS=7 timeout 150 python tools/conv_sweep.py > $O/sweep_persistent.txt 2> $O/err1.log || tail -5 $O/err1.log
cat $O/sweep_persistent.txt
timeout 150 python tools/roofline_ab.py "conv_fast=27" "conv_fast=59" "conv_fast=27" "conv_fast=59" --dump $O/launches > $O/roofline_ab.txt 2> $O/err3.log || tail -5 $O/err3.log
cat $O/roofline_ab.txt

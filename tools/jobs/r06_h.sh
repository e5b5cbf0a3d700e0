# round 6, job H: one RT-DETRv2 forward alone at batch 16 and 64: kernels (rocprofv3) and the per-launch conv table
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06h; rm -rf $O; mkdir -p $O
for b in 16 64; do
  timeout 200 python tools/rtdetr_profile.py --batch $b --dump > $O/rtdetr_b$b.txt 2> $O/rtdetr_b${b}_launches.txt; head -2 $O/rtdetr_b$b.txt
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b$b -o kt -- python $GRAFT_REPO_ROOT/tools/rtdetr_profile.py --batch $b --reps 10 > /dev/null 2> $O/kt_b$b.log)
  f=$(ls $O/kt_b$b/*/*kernel_stats.csv | head -1); head -30 $f | cut -c1-160
  find $O/kt_b$b -name "*kernel_trace.csv" -delete
done
grep -c . $O/rtdetr_b16_launches.txt; head -5 $O/rtdetr_b16_launches.txt

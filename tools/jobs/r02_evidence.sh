# evidence runs: recogniser alone on one page (kernel trace), DBNet MFMA-busy PMC pass, whole bench under kernel trace,
# wave-size variants of the bench
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r02e; rm -rf $O; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats -d $O/rec -o rec -- python tools/rec_only.py rec > $O/rec.log 2>&1 || tail -3 $O/rec.log
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/mfma -o m -- python bench.py --workload detector --steps 1 --warmup 1 --no-cpu-baseline > $O/mfma.json 2> $O/mfma.log || tail -3 $O/mfma.log
timeout 500 rocprofv3 --kernel-trace --stats -d $O/bench -o b -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_traced.json 2> $O/bench.log || tail -3 $O/bench.log
for wv in "16 2" "4 4"; do set -- $wv
  timeout 400 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --wave $1 --workers $2 > $O/wave$1_w$2.json 2>> $O/wave.log || tail -3 $O/wave.log
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r02e/*.json")):
    try:
        d=json.load(open(f)); r=d.get("roofline") or {}; print(f, d["value"], d["ms_per_step"], r.get("achieved"), r.get("conv_share_of_wall"))
    except Exception as e: print(f,"ERR",e)
PY
tail -2 $O/rec.log; du -sh $O

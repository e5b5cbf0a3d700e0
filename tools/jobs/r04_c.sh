# round 4, job C: where the time goes with the fp16 split as the models' default - per-launch dump of the serial pass,
# kernel stats of the timed job (all kernels), kernel stats of the default model set
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04c; rm -rf $O; mkdir -p $O
echo "== new tests"; timeout 600 python -m pytest tests/test_baseline_configs_gpu.py tests/test_conv_split_gpu.py -m gpu -q -x -k "reference or split" 2>&1 | grep -v "INFO\|^$" | tail -6
echo "== per-launch dump (analyzer, serial pass)"
YMK_DEBUG_OPTIONS="prof_dump=1" timeout 400 python bench.py --roofline-only --no-cpu-baseline > $O/roofline_only.json 2> $O/launches.txt || tail -5 $O/launches.txt
grep -c "ymk-prof" $O/launches.txt
python - <<'PY'
import re,collections
rows=[]
for l in open("gpurun_out/r04c/launches.txt"):
    m=re.search(r"\[ymk-prof\]\s+(\d+) (M=\s*\d+ Cin=\s*\d+ Cout=\s*\d+ k=\dx\d s=\d d=\d) tile=(\S+) ksplit=(\d+) grid=(\d+)\s+([\d.]+) us\s+([\d.]+) TFLOP", l)
    if m: rows.append((m.group(2), m.group(3), int(m.group(4)), float(m.group(6)), float(m.group(7))))
agg=collections.OrderedDict()
for d,t,ks,us,tf in rows:
    k=(d,t,ks); a=agg.setdefault(k,[0,0.0,0.0]); a[0]+=1; a[1]+=us; a[2]+=tf
tot=sum(a[1] for a in agg.values())
print("total us", tot, "launches", len(rows))
for k,a in sorted(agg.items(), key=lambda kv:-kv[1][1])[:70]:
    print(f"{k[0]} tile={k[1]:8s} ks={k[2]:3d} n={a[0]:4d} tot={a[1]:9.0f}us {100*a[1]/tot:5.1f}% avg={a[1]/a[0]:8.1f}us {a[2]/a[0]:6.1f}TF")
PY
echo "== kernel stats of the timed job (lite, f16 default)"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lite -o lite -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_lite.json 2> $O/err1.log || tail -5 $O/err1.log
find $O/prof_lite -name "*kernel_stats.csv" -exec cp {} $O/lite_kernel_stats.csv \; ; rm -rf $O/prof_lite
head -24 $O/lite_kernel_stats.csv | cut -c1-200
echo "== default model set under rocprofv3"
timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o default -- python bench.py --model-set default --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > $O/bench_default_set.json 2> $O/err5.log || tail -5 $O/err5.log
find $O/prof_default -name "*kernel_stats.csv" -exec cp {} $O/default_set_kernel_stats.csv \; ; rm -rf $O/prof_default
head -24 $O/default_set_kernel_stats.csv | cut -c1-200
cut -c1-200 $O/bench_default_set.json

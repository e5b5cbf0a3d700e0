# round 4, job W: per-layer table of the analyzer's conv launches against their binding roofs (prof_dump of the roofline pass)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r04w; rm -rf $O; mkdir -p $O
YMK_DEBUG_OPTIONS=prof_dump=1 timeout 300 python bench.py --roofline-only --no-cpu-baseline > $O/line.json 2> $O/dump.txt || tail -5 $O/dump.txt
grep -c "ymk-prof" $O/dump.txt
python tools/two_roof.py $O/dump.txt $O/two_roof_by_layer.md 3
grep "ymk-prof" $O/dump.txt | gzip > $O/dump.txt.gz; rm $O/dump.txt
head -30 $O/two_roof_by_layer.md | cut -c1-220

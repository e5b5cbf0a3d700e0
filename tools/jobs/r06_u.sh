# round 6, job U: evidence runs on the last tree - kernel statistics of the serial pass (what the pinned staging did to the copy
# launches), 48 more pages of the second layout head (figures, tables with spans) against the free-running oracle
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06u; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --roofline-only --no-cpu-baseline < /dev/null > $O/line_kt.json 2> $O/kt.log) || tail -5 $O/kt.log
python - $O/kt/kt_kernel_stats.csv < /dev/null <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows:
    if "copyBuffer" in r["Name"] or "fillBuffer" in r["Name"]:
        print(r["Name"][:40], r["Calls"], round(float(r["AverageNs"]) / 1e3, 2), "us", round(100 * float(r["TotalDurationNs"]) / tot, 2), "%")
PY
cp $O/kt/kt_kernel_stats.csv $O/serial_kernel_stats_last_tree.csv; rm -rf $O/kt
timeout 900 python tools/e2e_oracle_eval.py --pages 48 --first-seed 100 --lay-seed 1248 --out $O/e2e_second_layout_head_48_pages_seeds_100_147.json < /dev/null 2> $O/e2e.err | cut -c1-1100; echo "rc ${PIPESTATUS[0]}"; grep "^page" $O/e2e.err | grep -v "equal exact: equal" | head

# round 6, job D: end-to-end parity against the free-running CPU oracle (tools/e2e_oracle_eval.py), 16 pages, both arithmetic modes;
# the 48-page split-vs-exact leaf comparison on checkpoints that yield tables
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r06d; rm -rf $O; mkdir -p $O
echo "== e2e vs oracle, 16 pages"
timeout 900 python tools/e2e_oracle_eval.py --pages 16 --out $O/e2e_oracle_eval_16_pages.json 2> $O/e2e.err; echo "rc $?"; grep -v INFO $O/e2e.err | tail -25
echo "== split vs exact, 48 pages, all four nets"
ALL=1 CONTROL=1 timeout 600 python tools/split_eval_pages.py 48 > $O/split_eval_pages_all_four_nets.json 2> $O/split.err; echo "rc $?"; cat $O/split_eval_pages_all_four_nets.json | cut -c1-1500; tail -3 $O/split.err

# round 6, job P: more of the free-running comparison against the oracle: another layout head (figures, spans), split_text_across_cells, 32 more pages
# (second run: the two arms with the second layout head again, after the tool learned to recognise a tie at the encoder top-k cut)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06p; rm -rf $O; mkdir -p $O
run() { name=$1; shift; timeout 600 python tools/e2e_oracle_eval.py "$@" --out $O/e2e_$name.json < /dev/null 2> $O/$name.err | cut -c1-1200; echo "$name rc ${PIPESTATUS[0]}"; grep "^page" $O/$name.err | grep -v "equal exact: equal" | head -8; }
run lay_seed_1248_16_pages --pages 16 --lay-seed 1248

run split_text_across_cells_lay_seed_1248_8_pages --pages 8 --lay-seed 1248 --split-text-across-cells


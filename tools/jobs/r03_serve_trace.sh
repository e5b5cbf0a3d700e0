# where does a serve() job spend its time: new GPU tests, host stage trace, device kernel timeline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r03b; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_serving_gpu.py "tests/test_parseq_gpu.py::test_grouped_forward_with_a_repetition_stopped_row_in_a_short_group" \
   "tests/test_pipeline_gpu.py::test_degenerate_quad_gets_a_placeholder_and_the_outputs_stay_aligned" tests/test_parseq_gpu.py -m gpu -q -x 2>&1 | grep -v INFO | tail -15
timeout 300 python tools/serve_trace.py --steps 3 > $O/host_trace_f3.json 2> $O/host_trace.err || tail -5 $O/host_trace.err
YMK_SWITCH_INTERVAL=0.005 timeout 300 python tools/serve_trace.py --steps 3 > $O/host_trace_f3_sw5ms.json 2>> $O/host_trace.err
timeout 300 python tools/serve_trace.py --steps 3 --wave 16 --in-flight 2 > $O/host_trace_w16.json 2>> $O/host_trace.err
timeout 600 rocprofv3 --kernel-trace -d $O/kt -o kt -- python tools/serve_trace.py --steps 2 > $O/host_trace_rocprof.json 2> $O/rocprof.err || tail -5 $O/rocprof.err
python tools/gpu_timeline.py $O/kt $O/gpu_timeline.json > /dev/null 2>> $O/rocprof.err || tail -5 $O/rocprof.err
rm -rf $O/kt
for f in $O/*.json; do echo "== $f"; cat $f; echo; done

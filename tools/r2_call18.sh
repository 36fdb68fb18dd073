#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call18
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -q -x -s -k "window_prefix or deconv_flow_tail" > $out/tests_wp.log 2>&1 ; rc0=$?
grep -E "conv2:|conv3:|d/d window|passed|failed|Error" $out/tests_wp.log | head -20
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 60 $out/engine_step_trace.md
rm -rf $out/trace
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 12 $out/tests.log
exit $(( rc0 + rc1 ))

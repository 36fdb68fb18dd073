#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call14
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -q -x -k "correlation_on_planes or resample2d or engine or flow_head" > $out/tests.log 2>&1 ; rc0=$?
tail -n 6 $out/tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
tail -n 1 $out/bench.json | cut -c1-160
UFR_PF_MFMA=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_nopfmfma.json 2>$out/bench2.err
tail -n 1 $out/bench_nopfmfma.json | cut -c1-160
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 40 $out/engine_step_trace.md
rm -rf $out/trace
exit $rc1

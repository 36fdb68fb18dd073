#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call42
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_igemm_gpu.py tests/test_engine_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 3 $out/tests.log
[ $rc1 -ne 0 ] && tail -n 40 $out/tests.log && exit $rc1
for i in 1 2; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140; done
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 12 $out/engine_step_trace.md
rm -rf $out/trace

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call15
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -q -x > $out/tests.log 2>&1 ; rc0=$?
tail -n 6 $out/tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests2.log 2>&1 ; rc0=$?
tail -n 4 $out/tests2.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
tail -n 1 $out/bench.json | cut -c1-160
for d in 1 2 4 3 6 7; do
UFR_CORR_DEBUG=$d timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = [e for e in d['roofline']['kernels'] if 'corr_fwd_planes' in e['kernel']]
print('debug $d', d['ms_per_step'], k[0]['ms'] if k else None)
"
done
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 30 $out/engine_step_trace.md
rm -rf $out/trace
exit $rc1

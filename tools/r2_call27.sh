#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call28
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -q -x -s -k "window_prefix or golden or banded" > $out/tests_wp.log 2>&1 ; rc0=$?
grep -E "conv1:|conv2:|conv3:|d/d window|passed|failed|Error" $out/tests_wp.log | head -20
[ $rc0 -ne 0 ] && tail -n 30 $out/tests_wp.log && exit $rc0
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py tests/test_ops_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 3 $out/tests.log
for i in 1 2; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140; done
UFR_CONV1_IGEMM=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 24 $out/engine_step_trace.md
rm -rf $out/trace
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_call28/trace_bench.json").read().strip().splitlines()[-1])
for k in d["roofline"]["kernels"]:
    if "conv1" in k["kernel"] or "prefix" in k["kernel"] or "window" in k["kernel"]:
        print(f'{k["kernel"][:60]:60s} {k["ms"]:8.4f} ms {k.get("achieved")} TF')
PY
exit $rc1

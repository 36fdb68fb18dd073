#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call12
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -q -x -k "correlation_on_planes or resample2d or engine" > $out/tests.log 2>&1 ; rc0=$?
tail -n 6 $out/tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
tail -n 1 $out/bench.json | cut -c1-160
UFR_CORR_PLANES=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_nocorrplanes.json 2>$out/bench2.err
tail -n 1 $out/bench_nocorrplanes.json | cut -c1-160
timeout -k 10 300 python tools/bench_hbm_ops.py > $out/hbm_ops.jsonl 2>$out/hbm_ops.err
grep resample $out/hbm_ops.jsonl | cut -c1-200
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 24 $out/engine_step_trace.md
rm -rf $out/trace
timeout -k 10 600 python -m pytest tests/test_flownetc_gpu.py tests/test_cone_gpu.py tests/test_models_gpu.py -q -x > $out/tests2.log 2>&1 ; rc2=$?
tail -n 5 $out/tests2.log
exit $(( rc1 + rc2 ))

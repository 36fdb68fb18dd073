#!/bin/bash
# engine on the headline step: whole-step bench on/off, kernel trace, parity suites with the engine on
set -o pipefail
out=gpurun_out/r2_call6
mkdir -p $out
export TMPDIR=/tmp
UFR_ENGINE=1 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_engine.json 2>$out/bench_engine.err ; rc0=$?
tail -n 1 $out/bench_engine.json | cut -c1-400; tail -n 3 $out/bench_engine.err
UFR_ENGINE=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_torch.json 2>$out/bench_torch.err ; rc1=$?
tail -n 1 $out/bench_torch.json | cut -c1-300
(cd /tmp && UFR_ENGINE=1 timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err) ; rc2=$?
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 45 $out/engine_step_trace.md
rm -rf $out/trace
UFR_ENGINE=1 timeout -k 10 1500 python -m pytest tests/test_flownetc_gpu.py tests/test_cone_gpu.py tests/test_train_glue_gpu.py tests/test_placement_gpu.py -q -x > $out/tests_engine.log 2>&1 ; rc3=$?
tail -n 8 $out/tests_engine.log
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -q -s -k "c5" > $out/tests_c5.log 2>&1 ; rc4=$?
grep -n "C5\|passed\|failed" $out/tests_c5.log | tail
exit $(( rc0 + rc1 + rc3 + rc4 ))

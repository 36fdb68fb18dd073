#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call10
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_igemm_gpu.py tests/test_engine_gpu.py -q -x > $out/kernel_tests.log 2>&1 ; rc0=$?
tail -n 6 $out/kernel_tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_default.json 2>$out/bench_default.err ; rc1=$?
tail -n 1 $out/bench_default.json | cut -c1-160
UFR_ENGINE_PREFIX=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_noprefix.json 2>$out/bench_noprefix.err
tail -n 1 $out/bench_noprefix.json | cut -c1-160
UFR_IGEMM_XCD=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_noxcd.json 2>$out/bench_noxcd.err
tail -n 1 $out/bench_noxcd.json | cut -c1-160
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 40 $out/engine_step_trace.md
rm -rf $out/trace
timeout -k 10 900 python -m pytest tests/test_flownetc_gpu.py tests/test_cone_gpu.py tests/test_train_glue_gpu.py tests/test_placement_gpu.py tests/test_sharding_gpu.py -q -x > $out/tests.log 2>&1 ; rc2=$?
tail -n 6 $out/tests.log
exit $(( rc1 + rc2 ))

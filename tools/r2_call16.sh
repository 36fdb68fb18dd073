#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call16
mkdir -p $out
export TMPDIR=/tmp
for d in 0 1 2 4 6 7; do UFR_CORR_DEBUG=$d timeout -k 10 200 python tools/bench_corr_planes.py >> $out/corr_planes_experiments.jsonl 2>>$out/exp.err; done
cat $out/corr_planes_experiments.jsonl
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests.log 2>&1 ; rc0=$?
tail -n 6 $out/tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
tail -n 1 $out/bench.json | cut -c1-160
UFR_ENGINE_WINDOW=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_nowindow.json 2>$out/bench2.err
tail -n 1 $out/bench_nowindow.json | cut -c1-160
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 36 $out/engine_step_trace.md
rm -rf $out/trace
exit $rc1

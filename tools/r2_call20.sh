#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call20
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -q -x -k "correlation_on_planes" > $out/tests_cp.log 2>&1 ; rc0=$?
tail -n 3 $out/tests_cp.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 200 python tools/bench_corr_planes.py > $out/corr_planes.jsonl 2>$out/exp.err
cat $out/corr_planes.jsonl
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 3 $out/tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc2=$?
tail -n 1 $out/bench.json | cut -c1-160
exit $(( rc1 + rc2 ))

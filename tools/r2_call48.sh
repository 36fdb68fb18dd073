#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call48
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -q -x -k "correlation_on_planes" > $out/tests_cp.log 2>&1 ; rc0=$?
tail -n 3 $out/tests_cp.log
[ $rc0 -ne 0 ] && tail -n 30 $out/tests_cp.log && exit $rc0
echo R2; timeout -k 10 200 python tools/bench_corr_planes.py > $out/corr_planes_r2.jsonl 2>$out/exp.err; cat $out/corr_planes_r2.jsonl
echo K2; UFR_CORR_PLANES_R2=0 timeout -k 10 200 python tools/bench_corr_planes.py > $out/corr_planes_k2.jsonl 2>>$out/exp.err; cat $out/corr_planes_k2.jsonl
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 2 $out/tests.log
for i in 1 2; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140
UFR_CORR_PLANES_R2=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140; done
exit $rc1

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call30
mkdir -p $out
export TMPDIR=/tmp
UFR_CORR_PLANES_K2=0 timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -q -x -k "correlation_on_planes" > $out/tests_cp.log 2>&1 ; rc0=$?
tail -n 3 $out/tests_cp.log
[ $rc0 -ne 0 ] && tail -n 30 $out/tests_cp.log && exit $rc0
echo K1-deep; UFR_CORR_PLANES_K2=0 timeout -k 10 200 python tools/bench_corr_planes.py > $out/corr_planes_k1.jsonl 2>>$out/exp.err; cat $out/corr_planes_k1.jsonl
echo K2; timeout -k 10 200 python tools/bench_corr_planes.py > $out/corr_planes_k2.jsonl 2>$out/exp.err; cat $out/corr_planes_k2.jsonl
for i in 1 2; do UFR_CORR_PLANES_K2=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140; done
for i in 1 2; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140; done

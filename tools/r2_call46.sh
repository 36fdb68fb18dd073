#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call46
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 2 $out/tests.log
[ $rc1 -ne 0 ] && tail -n 40 $out/tests.log && exit $rc1
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140
UFR_IGEMM_M64_PREFIX=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140
done

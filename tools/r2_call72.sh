#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call72
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_igemm_gpu.py tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
for v in 1 0 1 0; do echo "UFR_IGEMM_SMALL_BATCH=$v"; UFR_IGEMM_SMALL_BATCH=$v timeout -k 10 300 python tools/bench_configs.py c2b1 --steps 50 2>/dev/null | cut -c1-200; done
timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-200

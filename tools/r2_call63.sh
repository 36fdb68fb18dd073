#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call63
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null > $out/bench_pipe.json; cut -c1-330 $out/bench_pipe.json
UFR_IGEMM_PIPE=0 timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null > $out/bench_nopipe.json; cut -c1-330 $out/bench_nopipe.json
timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-330

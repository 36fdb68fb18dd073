#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call54
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -q -x -k "corr_backward_window or attack_matches or banded" > $out/tests.log 2>&1 ; rc1=$?
tail -n 2 $out/tests.log
[ $rc1 -ne 0 ] && tail -n 40 $out/tests.log && exit $rc1
timeout -k 10 600 python tools/bench_configs.py c2b1 --steps 40 2>/dev/null

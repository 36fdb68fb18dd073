#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call4
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -q -x -s > $out/engine_tests.log 2>&1
rc1=$?
tail -n 30 $out/engine_tests.log
timeout -k 10 1200 python -m pytest tests/test_models_gpu.py -q -s -k "c5 or raft_gradient" > $out/tests.log 2>&1
rc=$?
grep -n "RAFT alt\|C5\|passed\|failed\|Error" $out/tests.log | tail -20
exit $(( rc1 + rc ))

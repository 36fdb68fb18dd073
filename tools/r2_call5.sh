#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call5
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py -q -x > $out/igemm_tests.log 2>&1 ; rc0=$?
tail -n 5 $out/igemm_tests.log
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py -q -s > $out/engine_tests.log 2>&1 ; rc1=$?
grep -n "engine\|flow:\|d/d\|passed\|failed\|Error\|assert" $out/engine_tests.log | tail -30
timeout -k 10 300 python tools/bench_igemm_layers.py > $out/igemm_layers.jsonl 2>&1 ; rc2=$?
cut -c1-200 $out/igemm_layers.jsonl
timeout -k 10 600 python -m pytest tests/test_models_gpu.py -q -s -k "c5" > $out/tests.log 2>&1 ; rc3=$?
grep -n "C5\|passed\|failed" $out/tests.log | tail
exit $(( rc0 + rc1 + rc2 + rc3 ))

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call3
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py -q -x -s > $out/igemm_tests.log 2>&1
rc1=$?
tail -n 15 $out/igemm_tests.log
[ $rc1 -ne 0 ] && exit $rc1
timeout -k 10 300 python tools/bench_igemm_layers.py > $out/igemm_layers.jsonl 2>&1 &&
timeout -k 10 1500 python -m pytest tests/test_models_gpu.py tests/test_train_glue_gpu.py tests/test_placement_gpu.py -q -s > $out/tests.log 2>&1
rc=$?
cat $out/igemm_layers.jsonl | cut -c1-220; grep -n "RAFT alt\|C5:\|passed\|failed\|Error" $out/tests.log | tail -20
exit $rc

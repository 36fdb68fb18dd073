#!/bin/bash
# round 2, GPU call 2: patch-coordinate step, tightened gates, new parity tests, RAFT float64 diagnosis
set -o pipefail
out=gpurun_out/r2_call2
mkdir -p $out
timeout -k 10 300 python tools/diag_raft_f64.py > $out/diag_raft.txt 2>&1 &&
MIOPEN_DEBUG_CONV_WINOGRAD=0 timeout -k 10 300 python tools/diag_raft_f64.py > $out/diag_raft_nowino.txt 2>&1 &&
timeout -k 10 1500 python -m pytest tests/test_flownetc_gpu.py tests/test_cone_gpu.py tests/test_sharding_gpu.py tests/test_models_gpu.py tests/test_train_glue_gpu.py tests/test_placement_gpu.py -q -x -s > $out/tests.log 2>&1
rc=$?
cat $out/diag_raft.txt | tail -12; tail -n 25 $out/tests.log
exit $rc

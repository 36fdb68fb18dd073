#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call59
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_models_gpu.py -q -x -k "FlowNet2 or flownet2 or flownets or c5 or C5 or predict_flow" > $out/tests.log 2>&1; rc=$?
tail -n 4 $out/tests.log
[ $rc -ne 0 ] && grep -E "^E |FAILED" $out/tests.log | head -10
timeout -k 10 600 python tools/bench_configs.py c5 --steps 20 2>/dev/null | cut -c1-200
UFR_ENGINE_FLOWNET2=0 timeout -k 10 600 python tools/bench_configs.py c5 --steps 20 2>/dev/null | cut -c1-200
exit $rc

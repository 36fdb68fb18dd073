#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call61
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_igemm_gpu.py -q -x -k "lds-dma-pipelined" > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
timeout -k 10 500 python tools/bench_igemm_layers.py --pipe 2>/dev/null | cut -c1-230 > $out/layers.jsonl
cat $out/layers.jsonl

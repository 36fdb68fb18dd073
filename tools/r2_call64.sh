#!/bin/bash
out=gpurun_out/r2_call64
mkdir -p $out
for cfg in "1 1" "1 0" "0 1" "0 0"; do
  set -- $cfg
  echo "== PIPE=$1 KORDER=$2" >> $out/chaos.log
  UFR_IGEMM_PIPE=$1 UFR_IGEMM_KORDER=$2 timeout -k 10 300 python -m pytest tests/test_cone_gpu.py -q -x -k "random_placements or cone_equals or windowed" 2>&1 | grep -E "AssertionError|passed|failed" | cut -c1-200 >> $out/chaos.log
done
cat $out/chaos.log

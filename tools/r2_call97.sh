#!/bin/bash
out=gpurun_out/r2_call97
mkdir -p $out
for cfg in "UFR_IGEMM_PIPE=5" "UFR_IGEMM_PIPE=7" "UFR_IGEMM_PIPE=0" "UFR_IGEMM_BUF=0" "UFR_IGEMM_PP64=1" "UFR_IGEMM_KORDER=0"; do
  echo "== $cfg" | tee -a $out/switch_tests.log
  env $cfg timeout -k 10 400 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py tests/test_igemm_gpu.py -q -x > $out/t.log 2>&1
  tail -n 1 $out/t.log | tee -a $out/switch_tests.log
done

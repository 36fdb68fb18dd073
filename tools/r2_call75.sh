#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call75
mkdir -p $out
for cfg in "UFR_IGEMM_PIPE=0" "UFR_IGEMM_KORDER=0" "UFR_IGEMM_PIPE=0 UFR_IGEMM_KORDER=0"; do
  echo "== $cfg" | tee -a $out/switch_tests.log
  env $cfg timeout -k 10 500 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py tests/test_igemm_gpu.py -q -x 2>&1 | tail -n 2 | tee -a $out/switch_tests.log
done
timeout -k 10 900 python tools/bench_configs.py c2 c2b1 c3 c3alt c4 c5 --steps 20 2>/dev/null | cut -c1-220 | tee $out/configs.jsonl

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call66
mkdir -p $out
for d in 0 1 2; do
  echo "== UFR_IGEMM_DBG=$d" >> $out/dbg.log
  UFR_IGEMM_DBG=$d timeout -k 10 200 python tools/bench_igemm_layers.py --pipe conv3_1 conv4_1 conv5_1 2>/dev/null | grep '"variant": 5' | cut -c1-60,110-230 >> $out/dbg.log || exit 1
done
cat $out/dbg.log

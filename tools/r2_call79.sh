#!/bin/bash
out=gpurun_out/r2_call79
mkdir -p $out
for d in 0 1 2; do
  echo "== UFR_IGEMM_DBG=$d" >> $out/dbg.log
  UFR_IGEMM_DBG=$d timeout -k 10 200 python -u tools/bench_igemm_layers.py --pp conv3_1 conv4_1 conv6_1 2>/dev/null | grep '"variant": 6' | cut -c1-75,125-230 >> $out/dbg.log || exit 1
done
cat $out/dbg.log

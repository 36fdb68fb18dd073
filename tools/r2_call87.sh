#!/bin/bash
out=gpurun_out/r2_call87
mkdir -p $out
for d in 0 1 2 3; do
  echo "== UFR_IGEMM_DBG=$d (1: no activation DMA after the first run, 2: no weight DMA after the first step, 3: neither)" >> $out/dbg.log
  UFR_IGEMM_DBG=$d timeout -k 10 200 python -u tools/bench_igemm_layers.py --pp3 conv3_1 conv4_1 2>/dev/null | grep '"variant": 7' | cut -c1-75,125-230 >> $out/dbg.log || exit 1
done
cat $out/dbg.log

#!/bin/bash
out=gpurun_out/r2_call91
mkdir -p $out
for d in 0 1 2 0 1 2; do
  echo "== UFR_IGEMM_PP_PRIO=$d (0: READ 2 / MFMA 0, 1: READ 0 / MFMA 2, 2: none)" >> $out/prio.log
  UFR_IGEMM_PP_PRIO=$d timeout -k 10 200 python -u tools/bench_igemm_layers.py --pp conv3_1 conv4_1 conv5_1 deconv3 2>/dev/null | grep '"variant": 6' | python -c "
import sys, json
best={}
for l in sys.stdin:
    d=json.loads(l); k=(d['layer'],d['dir']); best[k]=min(best.get(k,9),d['ms'])
print(best)" >> $out/prio.log || exit 1
done
cat $out/prio.log

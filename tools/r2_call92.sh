#!/bin/bash
out=gpurun_out/r2_call92
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py -q -k "ping-pong and not tap" > $out/tests.log 2>&1; rc=$?
tail -n 2 $out/tests.log; grep -E "^E  |FAILED" $out/tests.log | head -10
[ $rc -ne 0 ] && exit $rc
for d in 1 0 1 0; do
  echo "== UFR_IGEMM_PP_BUF=$d" >> $out/buf.log
  UFR_IGEMM_PP_BUF=$d timeout -k 10 200 python -u tools/bench_igemm_layers.py --pp conv3_1 conv4_1 conv5_1 deconv3 conv4 2>/dev/null | grep '"variant": 6' | python -c "
import sys, json
best={}
for l in sys.stdin:
    d=json.loads(l); k=(d['layer'],d['dir']); best[k]=min(best.get(k,9),d['ms'])
print(best)" >> $out/buf.log || exit 1
done
cat $out/buf.log

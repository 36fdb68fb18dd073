#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call88
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py -q -k "tap-reuse" > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
grep -E "^E  |FAILED" $out/tests.log | head -20
[ $rc -ne 0 ] && exit $rc
timeout -k 10 500 python -u tools/bench_igemm_layers.py --pp3 > $out/layers.jsonl 2>$out/err.log
python - <<'PY'
import json
best={}
for l in open('gpurun_out/r2_call88/layers.jsonl'):
    d=json.loads(l); k=(d['layer'],d['dir']); v=d['variant']
    best.setdefault(k,{}); cur=best[k].get(v,(9,0))
    if d['ms']<cur[0]: best[k][v]=(d['ms'],d['splitk'])
for k,v in best.items(): print(k,'tap-reuse',v.get(7),'ping-pong',v.get(6))
PY

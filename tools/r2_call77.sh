#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call77
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py -q -x -k "ping-pong" > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED|Error" $out/tests.log | head -10; exit $rc; }
timeout -k 10 500 python -u tools/bench_igemm_layers.py --pp > $out/layers.jsonl 2>$out/err.log
python - <<'PY'
import json
best={}
for l in open('gpurun_out/r2_call77/layers.jsonl'):
    d=json.loads(l); k=(d['layer'],d['dir']); v=d['variant']
    best.setdefault(k,{}); cur=best[k].get(v,(9,0))
    if d['ms']<cur[0]: best[k][v]=(d['ms'],d['splitk'])
for k,v in best.items(): print(k,'ping-pong',v.get(6),'pipelined',v.get(5))
PY

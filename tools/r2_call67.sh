#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call68
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_igemm_gpu.py -q -x -k "lds-dma-pipelined" > $out/tests.log 2>&1; rc=$?
tail -n 2 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
timeout -k 10 500 python tools/bench_igemm_layers.py --pipe 2>/dev/null | grep '"variant": 5' | cut -c1-60,110-230 > $out/layers.txt
python - <<'PY'
import json,re
best={}
for l in open('gpurun_out/r2_call68/layers.txt'):
    m=re.search(r'"layer": "(\w+)", "dir": "(\w+)".*"splitk": (\d+).*"ms": ([\d.]+)',l)
    k=(m.group(1),m.group(2),int(m.group(3))); best[k]=min(best.get(k,9),float(m.group(4)))
for k,v in best.items(): print(k,v)
PY
timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-200

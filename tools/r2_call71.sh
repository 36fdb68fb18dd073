#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call71
mkdir -p $out
timeout -k 10 500 python tools/bench_igemm_layers.py --pipe2 2>/dev/null | cut -c1-60,110-230 > $out/layers.txt
python - <<'PY'
import json,re
best={}
for l in open('gpurun_out/r2_call71/layers.txt'):
    m=re.search(r'"layer": "(\w+)", "dir": "(\w+)", "variant": (\d).*"splitk": (\d+).*"ms": ([\d.]+)',l)
    k=(m.group(1),m.group(2),int(m.group(4))); best.setdefault(k,{}); v=int(m.group(3)); best[k][v]=min(best[k].get(v,9),float(m.group(5)))
for k,v in best.items(): print(k,'v5',v[5],'v6 (prio + late A issue)',v[6])
PY
for v in 1 6 1 6 0; do echo "UFR_IGEMM_PIPE=$v"; UFR_IGEMM_PIPE=$v timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-200; done

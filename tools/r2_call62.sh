#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call62
mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_igemm_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
for k in 1 0; do
  echo "== UFR_IGEMM_KORDER=$k" >> $out/layers.txt
  UFR_IGEMM_KORDER=$k timeout -k 10 500 python tools/bench_igemm_layers.py --pipe 2>/dev/null | cut -c1-230 >> $out/layers.txt || exit 1
done
python - <<'PY'
import json
rows={}
k=None
for l in open('gpurun_out/r2_call62/layers.txt'):
    if l.startswith('=='): k=l.strip()[-1]; continue
    d=json.loads(l)
    key=(d['layer'],d['dir'],d['variant'],d['splitk'])
    rows.setdefault(key,{}).setdefault(k,[]).append(d['ms'])
for key,v in rows.items():
    print(key, 'korder1', min(v.get('1',[0])), 'korder0', min(v.get('0',[0])))
PY
for k in 1 0; do UFR_IGEMM_KORDER=$k timeout -k 10 600 python bench.py --steps 30 --warmup 5 2>/dev/null | cut -c1-400; done

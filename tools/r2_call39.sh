#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call39
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python tools/bench_igemm_layers.py --m64 > $out/layers_m64.jsonl 2>$out/layers.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r2_call39/layers_m64.jsonl")]
d={}
for r in rows: d.setdefault((r["layer"],r["dir"],r["splitk"]),{}).setdefault(r["variant"],[]).append(r["ms"])
for k,v in d.items():
    if 2 in v and 4 in v: print(f'{k[0]:8s} {k[1]} S={k[2]:2d}  v2 {v[2]}  v4 {v[4]}  {min(v[2])/min(v[4]):.2f}x')
PY

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call38
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_igemm_gpu.py -q -x > $out/tests_ig.log 2>&1 ; rc0=$?
tail -n 3 $out/tests_ig.log
[ $rc0 -ne 0 ] && tail -n 40 $out/tests_ig.log && exit $rc0
timeout -k 10 600 python tools/bench_igemm_layers.py --big > $out/layers_big.jsonl 2>$out/layers.err
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r2_call38/layers_big.jsonl")]
key=lambda r:(r["layer"],r["dir"],r["splitk"])
d={}
for r in rows: d.setdefault(key(r),{})[r["variant"]]=r
for k,v in d.items():
    if 2 in v and 3 in v: print(f'{k[0]:8s} {k[1]} S={k[2]:2d} M={v[2]["M"]:6d} N={v[2]["Npad"]:5d} K={v[2]["ktiles"]:4d}  v2 {v[2]["ms"]:.4f} ({v[2]["tflops"]:.0f})  v3 {v[3]["ms"]:.4f} ({v[3]["tflops"]:.0f})  {v[2]["ms"]/v[3]["ms"]:.2f}x')
PY

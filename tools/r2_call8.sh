#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call8
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py -q -x > $out/igemm_tests.log 2>&1 ; rc0=$?
tail -n 5 $out/igemm_tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 300 python tools/bench_igemm_layers.py --both > $out/igemm_layers.jsonl 2>&1
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r2_call8/igemm_layers.jsonl") if l.startswith("{")]
for v in (1,2):
    best={}
    for r in rows:
        if r["variant"]!=v: continue
        k=(r["layer"],r["dir"])
        if k not in best or r["ms"]<best[k]["ms"]: best[k]=r
    for d in ("fwd","bwd"):
        print("variant",v,d, "total ms", round(sum(x["ms"] for k,x in best.items() if k[1]==d),3), " ".join(f'{k[0]}:{x["ms"]:.3f}/{x["tflops"]:.0f}' for k,x in best.items() if k[1]==d))
PY
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -q -x > $out/engine_tests.log 2>&1 ; rc1=$?
tail -n 3 $out/engine_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_glds.json 2>$out/bench_glds.err ; rc2=$?
tail -n 1 $out/bench_glds.json | cut -c1-200
UFR_IGEMM=reg timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_reg.json 2>$out/bench_reg.err
tail -n 1 $out/bench_reg.json | cut -c1-200
exit $(( rc1 + rc2 ))

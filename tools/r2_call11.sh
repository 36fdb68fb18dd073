#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call11
mkdir -p $out
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py -q -x -k "flow_head" > $out/pf_tests.log 2>&1 ; rc0=$?
tail -n 3 $out/pf_tests.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
python - <<'PY'
import json
l=json.loads(open("gpurun_out/r2_call11/bench.json").read().strip().splitlines()[-1]); print(l["value"], l["ms_per_step"])
for k in l["roofline"]["kernels"]:
    if "flow_head" in k["kernel"] or "MIOpen" in k["kernel"]: print(k["kernel"], k["ms"], k.get("frac"))
PY
timeout -k 10 300 python tools/bench_hbm_ops.py > $out/hbm_ops.jsonl 2>$out/hbm_ops.err ; rc2=$?
cut -c1-230 $out/hbm_ops.jsonl; tail -n 3 $out/hbm_ops.err
timeout -k 10 600 python tools/bench_configs.py c3 c3alt c4 c5 --steps 10 > $out/configs.jsonl 2>$out/configs.err ; rc3=$?
cat $out/configs.jsonl
exit $(( rc0 + rc1 + rc2 + rc3 ))

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call45
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_default.json 2>/dev/null
UFR_IGEMM=m64 timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_m64.json 2>/dev/null
python - <<'PY'
import json
a = json.loads(open("gpurun_out/r2_call45/bench_default.json").read().strip().splitlines()[-1])
b = json.loads(open("gpurun_out/r2_call45/bench_m64.json").read().strip().splitlines()[-1])
print("step", a["ms_per_step"], b["ms_per_step"])
kb = {k["kernel"]: k for k in b["roofline"]["kernels"]}
for k in a["roofline"]["kernels"]:
    if "igemm" in k["kernel"]:
        o = kb.get(k["kernel"])
        print(f'{k["kernel"][:34]:34s} 128: {k["ms"]:.4f}  m64: {o["ms"]:.4f}  {k["ms"]/o["ms"]:.2f}x')
PY

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call19
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_call19/bench.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["value"], {k: d["roofline"][k] for k in ("achieved", "frac", "ms_per_iteration")})
for k in d["roofline"]["kernels"]:
    if "igemm" in k["kernel"]:
        print(f'{k["kernel"]:34s} {k["ms"]:8.4f} ms {k["achieved"]:7.1f} TF  {k["gflop"]:8.2f} GF')
PY
exit $rc1

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call33
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_igemm_gpu.py -q -x > $out/tests_ig.log 2>&1 ; rc0=$?
tail -n 3 $out/tests_ig.log
[ $rc0 -ne 0 ] && tail -n 40 $out/tests_ig.log && exit $rc0
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 3 $out/tests.log
for i in 1 2; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_$i.json 2>/dev/null; tail -n 1 $out/bench_$i.json | cut -c1-140; done
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r2_call33/bench_2.json").read().strip().splitlines()[-1])
for k in d["roofline"]["kernels"]:
    if "bwd" in k["kernel"] and "igemm" in k["kernel"]:
        print(f'{k["kernel"][:40]:40s} {k["ms"]:8.4f} ms {k.get("achieved")} TF')
PY
exit $rc1

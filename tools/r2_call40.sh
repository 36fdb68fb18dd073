#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call40
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_engine_gpu.py tests/test_cone_gpu.py -q -x > $out/tests.log 2>&1 ; rc1=$?
tail -n 3 $out/tests.log
[ $rc1 -ne 0 ] && tail -n 40 $out/tests.log && exit $rc1
for i in 1 2; do
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_on_$i.json 2>/dev/null; tail -n 1 $out/bench_on_$i.json | cut -c1-140
UFR_IGEMM_M64=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_off_$i.json 2>/dev/null; tail -n 1 $out/bench_off_$i.json | cut -c1-140
done
python - <<'PY'
import json
on = json.loads(open("gpurun_out/r2_call40/bench_on_2.json").read().strip().splitlines()[-1])
off = json.loads(open("gpurun_out/r2_call40/bench_off_2.json").read().strip().splitlines()[-1])
ko = {k["kernel"]: k for k in off["roofline"]["kernels"]}
for k in on["roofline"]["kernels"]:
    if any(n in k["kernel"] for n in ("conv4 bwd", "conv5 bwd", "conv6 bwd", "deconv3 bwd")):
        print(f'{k["kernel"][:36]:36s} m64 {k["ms"]:.4f}  128 {ko[k["kernel"]]["ms"]:.4f}')
PY

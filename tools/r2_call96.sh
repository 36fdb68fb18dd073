#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call96
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_igemm_gpu.py tests/test_engine_gpu.py tests/test_cone_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -n 2 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
for v in 1 0 1 0; do echo "UFR_IGEMM_BUF=$v" >> $out/bench.log; UFR_IGEMM_BUF=$v timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-frame 2>/dev/null > $out/bench.json; python -c "
import json; l=json.load(open('$out/bench.json')); print(l['ms_per_step'], l['roofline']['ms_per_iteration'], l['roofline']['frac'], [(k['kernel'][6:], k['ms']) for k in l['roofline']['kernels'] if 'deconv2 fwd' in k['kernel'] or 'conv1 fwd (pre' in k['kernel'] or 'conv2 fwd (win' in k['kernel']])" | tee -a $out/bench.log; done

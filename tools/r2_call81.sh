#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call81
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_igemm_gpu.py tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_flownetc_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -n 3 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/tests.log | head -10; exit $rc; }
for v in "0" "6" "0" "6"; do echo "UFR_CONV1_VARIANT=$v" >> $out/bench.log; UFR_CONV1_VARIANT=$v timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-frame 2>/dev/null > $out/bench_$v.json; python -c "
import json; l=json.load(open('$out/bench_$v.json')); print(l['ms_per_step'], l['roofline']['ms_per_iteration'], l['roofline']['frac'], [(k['kernel'][6:], k['ms']) for k in l['roofline']['kernels'] if 'conv1' in k['kernel'] or 'deconv2 fwd' in k['kernel'] or 'conv_redir' in k['kernel'] or 'conv2 bwd' in k['kernel']])" | tee -a $out/bench.log; done

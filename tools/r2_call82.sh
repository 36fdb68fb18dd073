#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call82
mkdir -p $out
for cfg in "UFR_IGEMM_PP64=1 UFR_CONV1_VARIANT=0" "UFR_IGEMM_PP64=0 UFR_CONV1_VARIANT=0" "UFR_IGEMM_PP64=1 UFR_CONV1_VARIANT=6" "UFR_IGEMM_PP64=1 UFR_CONV1_VARIANT=0" "UFR_IGEMM_PP64=0 UFR_CONV1_VARIANT=0" "UFR_IGEMM_PP64=1 UFR_CONV1_VARIANT=6"; do echo "$cfg" | tee -a $out/bench.log; env $cfg timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-frame 2>/dev/null > $out/bench.json; python -c "
import json; l=json.load(open('$out/bench.json')); print(l['ms_per_step'], l['roofline']['ms_per_iteration'], l['roofline']['frac'], [(k['kernel'][6:], k['ms']) for k in l['roofline']['kernels'] if 'conv1 fwd (pre' in k['kernel'] or 'deconv2 fwd' in k['kernel'] or 'conv2 bwd' in k['kernel']])" | tee -a $out/bench.log; done

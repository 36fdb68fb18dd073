#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call83
mkdir -p $out
for cfg in "UFR_SPLITK_MODEL=work" "UFR_SPLITK_MODEL=rounds" "UFR_SPLITK_MODEL=work" "UFR_SPLITK_MODEL=rounds"; do echo "$cfg" | tee -a $out/bench.log; env $cfg timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-frame 2>/dev/null > $out/bench.json; python -c "
import json; l=json.load(open('$out/bench.json')); print(l['ms_per_step'], l['roofline']['ms_per_iteration'], l['roofline']['frac'], [(k['kernel'][6:], k['ms']) for k in l['roofline']['kernels'] if ('conv4 bwd' in k['kernel'] or 'conv5 bwd' in k['kernel'] or 'conv6 bwd' in k['kernel'] or 'conv3 bwd' in k['kernel'] or 'conv2 bwd' in k['kernel'])])" | tee -a $out/bench.log; done

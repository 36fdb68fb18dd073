#!/bin/bash
set -o pipefail
for v in 1 0 1 0; do echo "UFR_IGEMM_M64_PREFIX=$v"; UFR_IGEMM_M64_PREFIX=$v timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-frame 2>/dev/null | python -c "
import sys, json
l=json.loads(sys.stdin.read()); print(l['ms_per_step'], [(k['kernel'][6:], k['ms']) for k in l['roofline']['kernels'] if 'prefix' in k['kernel'] or 'window' in k['kernel']])"; done

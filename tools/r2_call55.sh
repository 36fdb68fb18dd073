#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call55
mkdir -p $out
export TMPDIR=/tmp
for c in 22 11 12 21 22 11; do
  UFR_CORR_BWD_CFG=$c timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = [e for e in d['roofline']['kernels'] if 'corr_bwd_window' in e['kernel']]
print('cfg $c', d['ms_per_step'], k[0]['ms'])
"
done

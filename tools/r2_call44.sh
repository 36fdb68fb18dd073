#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call44
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -x -k "resample2d" > $out/tests.log 2>&1; tail -n 2 $out/tests.log
for m in full stream full stream; do
  echo "grid=$m"; UFR_RESAMPLE_GRID=$m timeout -k 10 300 python tools/bench_hbm_ops.py --resample-only 2>/dev/null | grep "resample2d_fwd" | cut -c1-150
done

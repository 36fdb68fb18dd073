#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call56
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1700 python -m pytest tests -m gpu -q -x -rs > $out/gpu_suite.log 2>&1 ; rc1=$?
tail -n 3 $out/gpu_suite.log
[ $rc1 -ne 0 ] && tail -n 40 $out/gpu_suite.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -n 2 $out/smoke.log
timeout -k 10 400 python bench.py --steps 20 --warmup 3 > $out/bench.json 2>/dev/null; tail -n 1 $out/bench.json | cut -c1-150
exit $rc1

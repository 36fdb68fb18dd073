#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call85
mkdir -p $out
timeout -k 10 1700 python -m pytest tests -m gpu -q -x > $out/gpu_suite.log 2>&1; rc=$?
tail -n 2 $out/gpu_suite.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED" $out/gpu_suite.log | head -10; exit $rc; }
timeout -k 10 600 python bench.py > $out/bench.json 2>$out/bench.err; python -c "
import json; l=json.load(open('$out/bench.json')); r=l['roofline']; print(l['value'], l['ms_per_step'], r['achieved'], r['frac'], r['traffic'], l['cpu_baseline']['value'])"
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 1

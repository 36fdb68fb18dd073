#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call101
mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_flownetc_gpu.py tests/test_engine_gpu.py tests/test_cone_gpu.py tests/test_models_gpu.py -q -x > $out/tests.log 2>&1; rc=$?
tail -n 2 $out/tests.log
[ $rc -ne 0 ] && { grep -E "^E |FAILED|Error" $out/tests.log | head -12; exit $rc; }
for v in 1 0 1 0; do echo "UFR_FUSED_LOSS=$v" >> $out/bench.log; UFR_FUSED_LOSS=$v timeout -k 10 600 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-full-frame 2>/dev/null > $out/bench.json; python -c "
import json; l=json.load(open('$out/bench.json')); print(l['ms_per_step'], l['roofline']['ms_per_iteration'])" | tee -a $out/bench.log; done

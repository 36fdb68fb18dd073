#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call43
mkdir -p $out
export TMPDIR=/tmp
rc=0
for v in big m64 reg; do
  UFR_IGEMM=$v timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_engine_gpu.py -q -x > $out/tests_$v.log 2>&1; r=$?
  echo "UFR_IGEMM=$v: $(tail -n 1 $out/tests_$v.log)"; [ $r -ne 0 ] && rc=1 && grep -E "^E |FAILED" $out/tests_$v.log | head -6
done
UFR_IGEMM_M64=1 timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_engine_gpu.py -q -x > $out/tests_m64sel.log 2>&1; echo "UFR_IGEMM_M64=1: $(tail -n 1 $out/tests_m64sel.log)"
exit $rc

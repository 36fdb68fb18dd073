#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call37
mkdir -p $out
export TMPDIR=/tmp
rc=0
for sw in UFR_CONV1_IGEMM UFR_ENGINE_PREFIX UFR_ENGINE; do
  env $sw=0 timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_engine_gpu.py -q -x > $out/tests_$sw.log 2>&1; r=$?
  echo "$sw=0: $(tail -n 1 $out/tests_$sw.log)"
  [ $r -ne 0 ] && rc=1 && grep -E "^E |FAILED|Error" $out/tests_$sw.log | head -8
done
timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_engine_gpu.py -q -x > $out/tests_default.log 2>&1; echo "default: $(tail -n 1 $out/tests_default.log)"
exit $rc

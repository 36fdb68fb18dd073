#!/bin/bash
# the A/B switches must keep working: the engine / cone suites with each fast path switched off in turn
set -o pipefail
out=gpurun_out/r2_call36
mkdir -p $out
export TMPDIR=/tmp
rc=0
for sw in UFR_ENGINE_WINDOW UFR_CONV1_IGEMM UFR_CORR_PLANES UFR_CORR_BWD_MFMA UFR_PF_MFMA UFR_DECONV_TAIL UFR_CORR_INCREMENTAL UFR_CORR_PLANES_K2 UFR_ENGINE_PREFIX UFR_ENGINE; do
  env $sw=0 timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_engine_gpu.py -q -x > $out/tests_$sw.log 2>&1; r=$?
  echo "$sw=0: $(tail -n 1 $out/tests_$sw.log)"
  [ $r -ne 0 ] && rc=1 && grep -E "^E |FAILED|Error" $out/tests_$sw.log | head -8
done
UFR_IGEMM=reg timeout -k 10 600 python -m pytest tests/test_cone_gpu.py tests/test_engine_gpu.py -q -x > $out/tests_reg.log 2>&1; echo "UFR_IGEMM=reg: $(tail -n 1 $out/tests_reg.log)"
exit $rc

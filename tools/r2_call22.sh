#!/bin/bash
# final-form evidence: full GPU suite, PMC traffic of the step, HBM-bound operator table
set -o pipefail
out=gpurun_out/r2_call22
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pf -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/pf.log 2>&1)
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pw -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/pw.log 2>&1)
ff=$(find $out/pf -name "*counter_collection.csv" | head -n 1); fw=$(find $out/pw -name "*counter_collection.csv" | head -n 1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_step_traffic.py $ff $fw 4 $out/r2_igemm_traffic.json $out > $out/r2_step_traffic.json 2>$out/pmc.err && head -c 1800 $out/r2_step_traffic.json && cat $out/r2_igemm_traffic.json $out/r2_corr_planes_traffic.json $out/r2_corr_window_traffic.json
rm -rf $out/pf $out/pw
timeout -k 10 300 python tools/bench_hbm_ops.py > $out/hbm_ops.jsonl 2>$out/hbm_ops.err
wc -l $out/hbm_ops.jsonl
timeout -k 10 1700 python -m pytest tests -m gpu -q -x > $out/gpu_suite.log 2>&1 ; rc1=$?
tail -n 8 $out/gpu_suite.log
exit $rc1

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call70
mkdir -p $out
export TMPDIR=/tmp
for v in 1 0; do
  (cd /tmp && UFR_IGEMM_PIPE=$v timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/tr$v -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-frame --steps 30 --warmup 5 > $GRAFT_REPO_ROOT/$out/bench$v.json 2>$GRAFT_REPO_ROOT/$out/err$v.log) || exit 1
  ft=$(find $out/tr$v -name "*kernel_trace.csv" | head -n 1)
  python tools/summarize_trace.py $ft 10 > $out/step_trace_pipe$v.md
  rm -rf $out/tr$v
  head -n 14 $out/step_trace_pipe$v.md | cut -c1-150
done

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call51
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/tools/bench_configs.py c2b1 --steps 40 > $GRAFT_REPO_ROOT/$out/c2b1.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
cat $out/c2b1.json
f=$(find $out/trace -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_stats.py $f 42 30 > $out/c2b1_stats.md 2>$out/summ.err && head -n 36 $out/c2b1_stats.md
rm -rf $out/trace

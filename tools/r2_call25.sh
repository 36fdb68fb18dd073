#!/bin/bash
# final-form evidence: the default bench line (cpu_baseline + full-frame side measurement) and the rocprofv3 --stats summary
set -o pipefail
out=gpurun_out/r2_call25
mkdir -p $out
export TMPDIR=/tmp
t0=$(date +%s)
timeout -k 10 1000 python bench.py > $out/bench_final.json 2>$out/bench_final.err ; rc0=$?
t1=$(date +%s); echo "bench.py wall $((t1 - t0)) s"
tail -n 1 $out/bench_final.json | python -c "
import sys, json
l=json.loads(sys.stdin.read()); print({k: l[k] for k in ('value','ms_per_step','vs_baseline','dtype')}); r=l['roofline']; print({k: r[k] for k in r if k not in ('kernels','step')}); print(r.get('step')); print(l.get('cpu_baseline')); print(l['config'].get('full_frame_attack_iters_per_s'))
"
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/stats_bench.json 2>$GRAFT_REPO_ROOT/$out/stats.err)
f=$(find $out/stats -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp $f $out/r2_final_kernel_stats.csv && python tools/summarize_stats.py $f 23 > $out/r2_final_kernel_stats.md 2>$out/summ.err && head -n 30 $out/r2_final_kernel_stats.md
ft=$(find $out/stats -name "*kernel_trace.csv" | head -n 1)
[ -n "$ft" ] && python tools/summarize_trace.py $ft 10 > $out/engine_step_trace.md 2>>$out/summ.err
rm -rf $out/stats
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -n 2 $out/smoke.log
exit $rc0

#!/bin/bash
# final-form evidence of round 2: full GPU suite, default bench line, rocprofv3 --stats, step trace, PMC traffic, HBM operator table
set -o pipefail
out=gpurun_out/r2_call102
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 1700 python -m pytest tests -m gpu -q -x -rs > $out/gpu_suite.log 2>&1 ; rc1=$?
tail -n 4 $out/gpu_suite.log
[ $rc1 -ne 0 ] && tail -n 40 $out/gpu_suite.log && exit $rc1
t0=$(date +%s)
timeout -k 10 1000 python bench.py > $out/bench_final.json 2>$out/bench_final.err ; rc0=$?
t1=$(date +%s); echo "bench.py wall $((t1 - t0)) s"
tail -n 1 $out/bench_final.json | python -c "
import sys, json
l=json.loads(sys.stdin.read()); print({k: l[k] for k in ('value','ms_per_step','vs_baseline','dtype')}); r=l['roofline']; print({k: r[k] for k in r if k not in ('kernels','step')}); print(r.get('step')); print(l.get('cpu_baseline')); print(l['config'].get('full_frame_attack_iters_per_s'))
"
(cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/stats_bench.json 2>$GRAFT_REPO_ROOT/$out/stats.err)
f=$(find $out/stats -name "*kernel_stats.csv" | head -n 1)
[ -n "$f" ] && cp $f $out/r2_final_kernel_stats.csv && python tools/summarize_stats.py $f 23 > $out/r2_final_kernel_stats.md 2>$out/summ.err && head -n 12 $out/r2_final_kernel_stats.md
ft=$(find $out/stats -name "*kernel_trace.csv" | head -n 1)
[ -n "$ft" ] && python tools/summarize_trace.py $ft 10 > $out/engine_step_trace.md 2>>$out/summ.err && head -n 30 $out/engine_step_trace.md
rm -rf $out/stats
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pf -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/pf.log 2>&1)
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pw -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/pw.log 2>&1)
ff=$(find $out/pf -name "*counter_collection.csv" | head -n 1); fw=$(find $out/pw -name "*counter_collection.csv" | head -n 1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_step_traffic.py $ff $fw 4 $out/r2_igemm_traffic.json $out > $out/r2_step_traffic.json 2>$out/pmc.err && head -c 600 $out/r2_step_traffic.json && cat $out/r2_igemm_traffic.json $out/r2_corr_planes_traffic.json $out/r2_corr_window_traffic.json
rm -rf $out/pf $out/pw
timeout -k 10 300 python tools/bench_hbm_ops.py > $out/hbm_ops.jsonl 2>$out/hbm_ops.err
wc -l $out/hbm_ops.jsonl
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; tail -n 2 $out/smoke.log
exit $rc0

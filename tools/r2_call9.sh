#!/bin/bash
# full GPU suite with the engine on + final-form bench line + traces + PMC traffic
set -o pipefail
out=gpurun_out/r2_call9
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python bench.py > $out/bench_full.json 2>$out/bench_full.err ; rc0=$?
tail -n 1 $out/bench_full.json | python -c "
import sys, json
l=json.loads(sys.stdin.read()); print({k: l[k] for k in ('value','ms_per_step')}); r=l['roofline']; print({k: r[k] for k in r if k not in ('kernels','step')}); print(r.get('step')); print(l.get('cpu_baseline')); print(l['config'].get('full_frame_attack_iters_per_s'))
for k in r['kernels']: print(k['kernel'], k['ms'], k.get('achieved'), k.get('frac'))
"
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pf -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/pf.log 2>&1)
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/pw -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame > $GRAFT_REPO_ROOT/$out/pw.log 2>&1)
ff=$(find $out/pf -name "*counter_collection.csv" | head -n 1); fw=$(find $out/pw -name "*counter_collection.csv" | head -n 1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_step_traffic.py $ff $fw 4 $out/r2_igemm_traffic.json > $out/r2_step_traffic.json 2>$out/pmc.err && head -c 1500 $out/r2_step_traffic.json && cat $out/r2_igemm_traffic.json
rm -rf $out/pf $out/pw
timeout -k 10 1700 python -m pytest tests -m gpu -q -x > $out/gpu_suite.log 2>&1 ; rc1=$?
tail -n 8 $out/gpu_suite.log
exit $(( rc0 + rc1 ))

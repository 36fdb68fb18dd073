#!/bin/bash
# One parametrised GPU call (replaces round 2's per-call scripts):
#   gpurun --timeout T -- 'bash tools/gpu_call.sh NAME STEP [STEP ...]'
# Steps run in order and stop at the first failure; outputs go to gpurun_out/NAME/.
#   suite            python -m pytest tests -m gpu -q -x -rs
#   tests:<a,b,..>   python -m pytest a b .. -q -x   (commas separate the words of a step's argument)
#   bench[:flags]    python bench.py <flags>
#   stats[:flags]    rocprofv3 --kernel-trace --stats of bench.py <flags> + the summaries under the same directory
#   trace:<tag>,<iters>,<script>,<args>   rocprofv3 --kernel-trace of a python tool + steady-state table (tools/summarize_trace.py;
#                    env:TRACE_MARKER=<kernel> names the last kernel of an iteration, default gate_kernel)
#   traffic:<tag>,<iters>,<script>,<args>   two --pmc passes (FETCH_SIZE / WRITE_SIZE) + tools/pmc_step_traffic.py
#   counters:<tag>,<C1+C2+..>,<kernel pattern>,<script>,<args>   one --pmc pass + tools/pmc_summary.py
#   configs:<list>   tools/bench_configs.py <list>
#   py:<script args> python <script args>
set -o pipefail
name=$1; shift
out=gpurun_out/$name
mkdir -p $out
export TMPDIR=/tmp
for step in "$@"; do
  kind=${step%%:*}; arg=""; [ "$kind" != "$step" ] && arg=${step#*:}
  arg=${arg//,/ }                       # commas separate the words of one step's argument
  echo "== $step"
  case $kind in
    suite)   timeout -k 10 1500 python -m pytest tests -m gpu -q -x -rs > $out/gpu_suite.log 2>&1; rc=$?; tail -n 6 $out/gpu_suite.log
             [ $rc -ne 0 ] && { grep -E "^E |FAILED|Error" $out/gpu_suite.log | head -n 20; exit $rc; } ;;
    tests)   nt=$((${nt:-0} + 1)); tl=$out/tests$([ $nt -gt 1 ] && echo _$nt).log     # a second tests step of a call: tests_2.log
             timeout -k 10 1100 python -m pytest $arg -q -x > $tl 2>&1; rc=$?; tail -n 4 $tl
             [ $rc -ne 0 ] && { grep -E "^E |FAILED|Error" $tl | head -n 30; exit $rc; } ;;
    bench)   nb=$((${nb:-0} + 1)); bj=$out/bench$([ $nb -gt 1 ] && echo _$nb).json      # a second bench step of a call: bench_2.json
             timeout -k 10 900 python bench.py $arg > $bj 2> $out/bench.err; rc=$?
             [ $rc -ne 0 ] && { tail -n 20 $out/bench.err; exit $rc; }
             python tools/summarize_bench.py $bj | head -n ${BENCH_LINES:-400} ;;
    stats)   (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out/stats -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-full-frame --no-other-configs $arg > $GRAFT_REPO_ROOT/$out/stats_bench.json 2> $GRAFT_REPO_ROOT/$out/stats.err); rc=$?
             [ $rc -ne 0 ] && { tail -n 20 $out/stats.err; exit $rc; }
             f=$(find $out/stats -name "*kernel_stats.csv" | head -n 1); ft=$(find $out/stats -name "*kernel_trace.csv" | head -n 1)
             [ -n "$f" ] && python tools/summarize_stats.py $f 25 > $out/kernel_stats.md && head -n 14 $out/kernel_stats.md
             [ -n "$ft" ] && python tools/summarize_trace.py $ft 10 > $out/step_trace.md && sed -n "1,/^last iteration/p" $out/step_trace.md | head -n 30
             rm -rf $out/stats ;;
    trace)   # trace:<tag>,<iters>,<script>,<args...>: rocprofv3 --kernel-trace of `python <script> <args>`, steady-state table
             set -- $arg; tag=$1; iters=$2; shift 2
             (cd /tmp && timeout -k 10 900 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace_$tag -- python $GRAFT_REPO_ROOT/$1 "${@:2}" > $GRAFT_REPO_ROOT/$out/trace_$tag.log 2>&1); rc=$?
             [ $rc -ne 0 ] && { tail -n 20 $out/trace_$tag.log; exit $rc; }
             ft=$(find $out/trace_$tag -name "*kernel_trace.csv" | head -n 1)
             [ -n "$ft" ] && python tools/summarize_trace.py $ft $iters ${TRACE_MARKER:-gate_kernel} > $out/${tag}_step_trace.md && sed -n "1,/^last iteration/p" $out/${tag}_step_trace.md | head -n 45
             rm -rf $out/trace_$tag ;;
    traffic) # traffic:<tag>,<iters>,<script>,<args...>: HBM bytes per iteration from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
             # each with --kernel-trace only), summed by tools/pmc_step_traffic.py (env:TRACE_MARKER as for trace)
             set -- $arg; tag=$1; iters=$2; shift 2
             for ctr in FETCH_SIZE WRITE_SIZE; do
               (cd /tmp && timeout -k 10 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d /tmp/pmc_${tag}_$ctr -- python $GRAFT_REPO_ROOT/$1 "${@:2}" > $GRAFT_REPO_ROOT/$out/pmc_${tag}_$ctr.log 2>&1); rc=$?
               [ $rc -ne 0 ] && { tail -n 20 $out/pmc_${tag}_$ctr.log; exit $rc; }
             done
             ff=$(find /tmp/pmc_${tag}_FETCH_SIZE -name "*counter_collection.csv" | head -n 1); fw=$(find /tmp/pmc_${tag}_WRITE_SIZE -name "*counter_collection.csv" | head -n 1)
             python tools/pmc_step_traffic.py $ff $fw $iters $out/${tag}_igemm_traffic.json $out ${TRACE_MARKER:-gate_kernel} > $out/${tag}_step_traffic.json; rc=$?
             [ $rc -ne 0 ] && exit $rc
             head -n 12 $out/${tag}_step_traffic.json; cat $out/${tag}_igemm_traffic.json ;;
    counters) # counters:<tag>,<COUNTER+COUNTER+...>,<kernel pattern>,<script>,<args...>: one rocprofv3 --pmc pass (with --kernel-trace only),
             # per-kernel averages by tools/pmc_summary.py
             set -- $arg; tag=$1; ctrs=${2//+/ }; pat=$3; shift 3
             (cd /tmp && timeout -k 10 900 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d /tmp/ctr_$tag -- python $GRAFT_REPO_ROOT/$1 "${@:2}" > $GRAFT_REPO_ROOT/$out/ctr_$tag.log 2>&1); rc=$?
             [ $rc -ne 0 ] && { tail -n 20 $out/ctr_$tag.log; exit $rc; }
             fc=$(find /tmp/ctr_$tag -name "*counter_collection.csv" | head -n 1)
             python tools/pmc_summary.py $fc $pat > $out/${tag}_counters.txt; rc=$?
             [ $rc -ne 0 ] && exit $rc
             cat $out/${tag}_counters.txt ;;
    configs) nc=$((${nc:-0} + 1)); cl=$out/configs$([ $nc -gt 1 ] && echo _$nc).log     # a second configs step of a call: configs_2.log
             timeout -k 10 1100 python tools/bench_configs.py $arg > $cl 2>&1; rc=$?; tail -n 12 $cl
             [ $rc -ne 0 ] && exit $rc ;;
    env)     export $arg ;;                      # env:NAME=VALUE for the steps behind it
    py)      np=$((${np:-0} + 1)); pl=$out/py$([ $np -gt 1 ] && echo _$np).log          # a second py step of a call: py_2.log
             timeout -k 10 1100 python $arg > $pl 2>&1; rc=$?; tail -n 40 $pl
             [ $rc -ne 0 ] && exit $rc ;;
    *)       echo "unknown step $step"; exit 2 ;;
  esac
done
exit 0

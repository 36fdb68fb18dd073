#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call13
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -q -x -k "correlation_on_planes or resample2d or engine" > $out/tests.log 2>&1 ; rc0=$?
tail -n 6 $out/tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
tail -n 1 $out/bench.json | cut -c1-160
timeout -k 10 300 python tools/bench_hbm_ops.py --resample-only > $out/hbm_resample_lds.jsonl 2>$out/hbm_ops.err
UFR_RESAMPLE_LDS=0 timeout -k 10 300 python tools/bench_hbm_ops.py --resample-only > $out/hbm_resample_direct.jsonl 2>>$out/hbm_ops.err
echo LDS; grep resample $out/hbm_resample_lds.jsonl | cut -c1-150
echo DIRECT; grep resample $out/hbm_resample_direct.jsonl | cut -c1-150
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 24 $out/engine_step_trace.md
rm -rf $out/trace
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc1 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/pmc1.log 2>&1)
f=$(find $out/pmc1 -name "*counter_collection.csv" | head -n 1)
if [ -n "$f" ]; then for k in flow_head_planes_fwd corr_fwd_planes corr_bwd_window flow_head_planes_bwd; do python tools/pmc_summary.py $f $k >> $out/small_kernels_pmc1.txt; done; cat $out/small_kernels_pmc1.txt; fi
rm -rf $out/pmc1
(cd /tmp && timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc2 -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/pmc2.log 2>&1)
f=$(find $out/pmc2 -name "*counter_collection.csv" | head -n 1)
if [ -n "$f" ]; then for k in flow_head_planes_fwd corr_fwd_planes corr_bwd_window flow_head_planes_bwd; do python tools/pmc_summary.py $f $k >> $out/small_kernels_pmc2.txt; done; cat $out/small_kernels_pmc2.txt; fi
rm -rf $out/pmc2
exit $rc1

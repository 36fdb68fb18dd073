#!/bin/bash
# engine default on: kernel tests, layer bench, whole-step bench + trace, PMC of the igemm on conv3_1
set -o pipefail
out=gpurun_out/r2_call7
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_igemm_gpu.py tests/test_engine_gpu.py -q -x > $out/kernel_tests.log 2>&1 ; rc0=$?
tail -n 5 $out/kernel_tests.log
timeout -k 10 300 python tools/bench_igemm_layers.py > $out/igemm_layers.jsonl 2>&1
python - <<'PY'
import json
rows=[json.loads(l) for l in open("gpurun_out/r2_call7/igemm_layers.jsonl") if l.startswith("{")]
best={}
for r in rows:
    k=(r["layer"],r["dir"])
    if k not in best or r["ms"]<best[k]["ms"]: best[k]=r
for d in ("fwd","bwd"):
    print(d, "total ms", round(sum(v["ms"] for k,v in best.items() if k[1]==d),3), " ".join(f'{k[0]}:{v["ms"]:.3f}/{v["tflops"]:.0f}' for k,v in best.items() if k[1]==d))
PY
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_engine.json 2>$out/bench_engine.err ; rc1=$?
tail -n 1 $out/bench_engine.json | cut -c1-200
(cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$out/trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $GRAFT_REPO_ROOT/$out/trace_bench.json 2>$GRAFT_REPO_ROOT/$out/trace.err)
f=$(find $out/trace -name "*kernel_trace.csv" | head -n 1)
[ -n "$f" ] && python tools/summarize_trace.py $f 10 > $out/engine_step_trace.md 2>$out/summ.err && head -n 30 $out/engine_step_trace.md
rm -rf $out/trace
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_igemm_layers.py conv3_1 > $GRAFT_REPO_ROOT/$out/pmc1.log 2>&1)
f=$(find $out/pmc1 -name "*counter_collection.csv" | head -n 1)
[ -n "$f" ] && python tools/pmc_summary.py $f igemm_kernel > $out/igemm_pmc1.txt && cat $out/igemm_pmc1.txt
rm -rf $out/pmc1
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_igemm_layers.py conv3_1 > $GRAFT_REPO_ROOT/$out/pmc2.log 2>&1)
f=$(find $out/pmc2 -name "*counter_collection.csv" | head -n 1)
[ -n "$f" ] && python tools/pmc_summary.py $f igemm_kernel > $out/igemm_pmc2.txt && cat $out/igemm_pmc2.txt
rm -rf $out/pmc2
exit $(( rc0 + rc1 ))

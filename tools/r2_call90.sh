#!/bin/bash
# PMC counters of the igemm forms on conv3_1 / conv4_1 (per-layer bench: variants 6 = ping-pong, 5 = pipelined, 2 = single-stage)
out=gpurun_out/r2_call90
mkdir -p $out
export TMPDIR=/tmp
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc1 -- python $GRAFT_REPO_ROOT/tools/bench_igemm_layers.py --pp conv3_1 conv4_1 > $GRAFT_REPO_ROOT/$out/pmc1.log 2>&1)
f=$(find $out/pmc1 -name "*counter_collection.csv" | head -n 1)
[ -n "$f" ] && for k in igemm_pp_kernel "igemm_glds_kernel<128, 128, true>"; do python tools/pmc_summary.py $f "$k" >> $out/igemm_v6_pmc1.txt; done
rm -rf $out/pmc1
(cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_MFMA --output-format csv -d $GRAFT_REPO_ROOT/$out/pmc2 -- python $GRAFT_REPO_ROOT/tools/bench_igemm_layers.py --pp conv3_1 conv4_1 > $GRAFT_REPO_ROOT/$out/pmc2.log 2>&1)
f=$(find $out/pmc2 -name "*counter_collection.csv" | head -n 1)
[ -n "$f" ] && for k in igemm_pp_kernel "igemm_glds_kernel<128, 128, true>"; do python tools/pmc_summary.py $f "$k" >> $out/igemm_v6_pmc2.txt; done
rm -rf $out/pmc2
cat $out/igemm_v6_pmc1.txt $out/igemm_v6_pmc2.txt
tail -n 3 $out/pmc2.log | cut -c1-200

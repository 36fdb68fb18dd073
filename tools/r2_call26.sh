#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call26
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests/test_engine_gpu.py tests/test_ops_gpu.py -q -x -k "correlation_on_planes or resample2d" > $out/tests.log 2>&1 ; rc0=$?
tail -n 3 $out/tests.log
[ $rc0 -ne 0 ] && exit $rc0
timeout -k 10 200 python tools/bench_corr_planes.py > $out/corr_planes.jsonl 2>$out/exp.err; cat $out/corr_planes.jsonl
for i in 1 2; do timeout -k 10 300 python tools/bench_hbm_ops.py --resample-only > $out/hbm_resample_$i.jsonl 2>>$out/hbm.err; grep resample $out/hbm_resample_$i.jsonl | cut -c1-140; done
UFR_RESAMPLE_LDS=0 timeout -k 10 300 python tools/bench_hbm_ops.py --resample-only > $out/hbm_resample_direct.jsonl 2>>$out/hbm.err; grep resample $out/hbm_resample_direct.jsonl | cut -c1-140
for i in 1 2; do timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline 2>/dev/null | tail -n 1 | cut -c1-140; done

#!/bin/bash
# First GPU call of round 2: everything written after round 1's GPU minutes ran out, in one go.
#   gpurun --timeout 1100 -- 'bash tools/round2_first_call.sh'
# Outputs under gpurun_out/r2_first/.  Steps are joined with && so that nothing runs after a failure.
set -o pipefail
out=gpurun_out/r2_first
mkdir -p $out
# 1. the gated tests: wide-tile kernel == measured kernel bit for bit, layout passes, conv_leaky through the split kernels
UFR_EXPERIMENTAL=1 timeout -k 10 200 python -m pytest tests/test_split_gemm_gpu.py -q > $out/tests.log 2>&1 &&
# 2. layer timings incl. the wide tile (skip MIOpen's find: round 1's numbers are in profiles/r1_split_conv_v1.jsonl)
UFR_EXPERIMENTAL=1 UFR_SKIP_MIOPEN=1 timeout -k 10 120 python tools/microbench_split_conv.py > $out/layers.jsonl 2>&1 &&
# 2b. the same with the XCD-aware tile order (read once per process, hence a second run)
UFR_SPLIT_XCD=1 UFR_EXPERIMENTAL=1 UFR_SKIP_MIOPEN=1 timeout -k 10 120 python tools/microbench_split_conv.py > $out/layers_xcd.jsonl 2>&1 &&
UFR_SPLIT_XCD=1 UFR_EXPERIMENTAL=1 timeout -k 10 200 python -m pytest tests/test_split_gemm_gpu.py -q > $out/tests_xcd.log 2>&1 &&
# 3. whole steps with the opt-in wiring: headline (bands on), then FlowNet2's universal step (no bands)
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_off.json 2>$out/bench_off.err &&
UFR_SPLIT_CONV=6 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_split6.json 2>$out/bench_split6.err &&
UFR_SPLIT_CONV=3 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_split3.json 2>$out/bench_split3.err &&
# 4. every frozen conv / deconv block on the split kernels (strided and transposed variants included)
UFR_EXPERIMENTAL=1 UFR_SPLIT_CONV=6 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_all6.json 2>$out/bench_all6.err
rc=$?
tail -n 3 $out/tests.log; tail -n 1 $out/tests_xcd.log; tail -n 4 $out/layers.jsonl | cut -c1-400; tail -n 4 $out/layers_xcd.jsonl | cut -c1-400; for f in $out/bench_*.json; do echo $f; tail -n 1 $f | cut -c1-300; done
exit $rc

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call50
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 900 python tools/bench_configs.py c2 c2b1 c3 c3alt c4 c5 --steps 10 > $out/configs.jsonl 2>$out/configs.err
cat $out/configs.jsonl

#!/bin/bash
out=gpurun_out/r2_call76
mkdir -p $out
for c in c2 c2b1 c5 c4 c3 c3alt; do
  timeout -k 10 400 python -u tools/bench_configs.py $c --steps 10 >> $out/configs.jsonl 2>>$out/err.log || exit 1
  tail -n 1 $out/configs.jsonl | cut -c1-220
done

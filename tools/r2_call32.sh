#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call32
mkdir -p $out
export TMPDIR=/tmp
UFR_DIST_BACKEND=gloo timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 > $out/bench_2ranks_gloo.json 2>$out/bench_2ranks.err ; rc=$?
tail -n 1 $out/bench_2ranks_gloo.json | cut -c1-400
[ $rc -ne 0 ] && tail -n 30 $out/bench_2ranks.err
timeout -k 10 600 python -m pytest tests/test_sharding_gpu.py -q -x > $out/tests.log 2>&1; tail -n 3 $out/tests.log
exit $rc

#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call17
mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_engine_gpu.py -q -x -s -k "window_prefix or deconv_flow_tail or engine_head" > $out/tests_wp.log 2>&1 ; rc0=$?
grep -E "conv2:|conv3:|d/d window|passed|failed|Error" $out/tests_wp.log | head -20
for d in 0 1 2 4 7; do UFR_CORR_DEBUG=$d timeout -k 10 200 python tools/bench_corr_planes.py >> $out/corr_planes_experiments.jsonl 2>>$out/exp.err; done
cat $out/corr_planes_experiments.jsonl
timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench.json 2>$out/bench.err ; rc1=$?
tail -n 1 $out/bench.json | cut -c1-160
UFR_ENGINE_WINDOW=0 timeout -k 10 400 python bench.py --steps 20 --warmup 3 --no-full-frame --no-cpu-baseline > $out/bench_nowindow.json 2>$out/bench2.err
tail -n 1 $out/bench_nowindow.json | cut -c1-160
python - <<'PY'
import json
for f in ("bench.json", "bench_nowindow.json"):
    try:
        d = json.loads(open("gpurun_out/r2_call17/" + f).read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], [(k["kernel"][6:], k["ms"], k["achieved"]) for k in d["roofline"]["kernels"] if "bwd" in k["kernel"] and "igemm" in k["kernel"]])
    except Exception as e:
        print(f, "unreadable", e)
PY
exit $rc0

#!/bin/bash
# Build everything in-tree, then send the tree to the GPU box:  tools/gpu.sh TIMEOUT NAME STEP [STEP ...]
# (a stale libufr_hip.so travelling with new tests cost round 3 its first call)
set -e
cd "$(dirname "$0")/.."
make -s -j8 -C understanding_flow_robustness_amd/csrc
md5sum understanding_flow_robustness_amd/lib/libufr_hip.so          # (which library travels; its objects embed the checksums of their sources:
mkdir -p gpurun_out                                                  #  `ufr_build_manifest`, verified by _lib.py at load on the box)
md5sum understanding_flow_robustness_amd/csrc/build/*.o > gpurun_out/last_build_objects.md5
make -s -C oracle all >/dev/null
t=$1; shift
exec /usr/local/graft/bin/gpurun --timeout $t -- "bash tools/gpu_call.sh $*"

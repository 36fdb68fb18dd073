#!/bin/bash
# flakiness screen: the whole GPU suite twice more (new kernels: counted-vmcnt correlation, fused window adjoint, per-phase split-K)
set -o pipefail
out=gpurun_out/r2_call47
mkdir -p $out
export TMPDIR=/tmp
rc=0
for i in 1 2; do
  timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $out/suite_$i.log 2>&1; r=$?
  echo "run $i: $(tail -n 1 $out/suite_$i.log)"; [ $r -ne 0 ] && rc=1 && grep -E "^E |FAILED" $out/suite_$i.log | head -8
done
# the correlation kernels' determinism: 20 launches, bit-identical outputs
python - <<'PY'
import torch, sys
sys.path.insert(0, ".")
from understanding_flow_robustness_amd import _lib as L, igemm as ig
lib = L.lib(); B, H, W = 8, 48, 160
g = torch.Generator().manual_seed(0)
f1 = ig.Planes(B, H, W, 8, "cuda:0").load_nchw(torch.randn(B, 256, H, W, generator=g).cuda(), 0)
f2 = ig.Planes(B, H, W, 8, "cuda:0").load_nchw(torch.randn(B, 256, H, W, generator=g).cuda(), 0)
ref = None
for i in range(20):
    out = ig.Planes(B, H, W, 15, "cuda:0")
    L.check(lib.ufr_corr_forward_planes(L.ptr(f1.t), L.ptr(f2.t), f1.plane_stride, L.ptr(out.t), out.plane_stride, 1, B, 256, H, W, 21, 2, 1.0 / 256.0, 0.1, L.stream()))
    torch.cuda.synchronize()
    if ref is None: ref = out.t.clone()
    elif not torch.equal(ref, out.t): print("MISMATCH at launch", i); sys.exit(1)
print("correlation planes: 20 launches bit-identical")
PY
exit $rc

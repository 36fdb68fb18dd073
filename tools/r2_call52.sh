#!/bin/bash
set -o pipefail
out=gpurun_out/r2_call52
mkdir -p $out
export TMPDIR=/tmp
for v in glds m64 glds m64; do echo "UFR_IGEMM=$v"; UFR_IGEMM=$v timeout -k 10 300 python - <<'PY'
import json, sys, os, time
sys.path.insert(0, ".")
import torch
from argparse import Namespace
from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
DEV = "cuda:0"
for B in (1, 2):
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1000.0, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    g = torch.Generator().manual_seed(0)
    H, W = 384, 1280
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    if B > 1:
        mask = torch.ones(1, 3, 51, 51, device=DEV); patch = torch.rand(1, 3, 51, 51, generator=g).to(DEV); placed = dict(origins=[(100, 600)] * B)
    else:
        mask = torch.zeros(B, 3, H, W, device=DEV); mask[:, :, 100:151, 600:651] = 1; patch = torch.rand(1, 3, H, W, generator=g).to(DEV); placed = {}
    target = torch.randn(B, 2, H, W, generator=g).to(DEV)
    step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51) if B > 1 else None)
    step.load(tgt, ref, patch, mask, patch, target, **placed)
    step.run(0); step.enqueue(2); torch.cuda.synchronize()
    t0 = time.perf_counter(); step.enqueue(40); torch.cuda.synchronize()
    print(f"  B={B}: {(time.perf_counter() - t0) * 1e3 / 40:.3f} ms / iteration (steady state)")
PY
done

"""RAFT image-gradient conditioning, with a float64 truth (VERDICT r1, item 1b / ADVICE: the 5e-2 gate).

    truth  = oracle.flow_oracle.raft_forward in float64 on the CPU (the restatement is pinned to the reference in fp32)
    cpu32  = the reference's own fp32 result (golden raft_128x192)
    A      = the product (HIP lookup / GRU / convex upsampling kernels + MIOpen convolutions), fp32, this GPU
    B      = the pure-torch spelling (oracle on HIP tensors: grid_sample lookup, torch GRU, unfold), fp32, this GPU
    C      = B in float64 on this GPU
Prints max-abs errors relative to max |truth| for the flow, both image gradients, and A vs B directly.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from argparse import Namespace

import torch
import torch.nn.functional as F

from conftest import load_golden, t
from oracle import flow_oracle as fo
from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow

DEV = "cuda:0"
z = load_golden("raft_128x192")


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / b.double().abs().max())


def oracle_run(sd, dev, dt, alternate=False):
    sdd = {k: (v.to(dev, dt) if v.is_floating_point() else v.to(dev)) for k, v in sd.items()}
    x1, x2 = t(z["x1"]).to(dev, dt).requires_grad_(True), t(z["x2"]).to(dev, dt).requires_grad_(True)
    flow = fo.raft_forward(sdd, x1 * 255.0, x2 * 255.0)[1]
    loss = fo.flow_loss(flow, t(z["target"]).to(dev, dt))
    g1, g2 = torch.autograd.grad(loss, (x1, x2))
    return flow.detach(), g1, g2


def product_run(net, args):
    x1, x2 = t(z["x1"], DEV).requires_grad_(True), t(z["x2"], DEV).requires_grad_(True)
    flow = predict_flow(net, None, x1, x2, args)
    loss = (1 - F.cosine_similarity(flow, t(z["target"], DEV))).mean()
    g1, g2 = torch.autograd.grad(loss, (x1, x2))
    return flow.detach(), g1, g2


def main():
    print("MIOPEN_DEBUG_CONV_WINOGRAD =", os.environ.get("MIOPEN_DEBUG_CONV_WINOGRAD"), " cudnn.benchmark =",
          torch.backends.cudnn.benchmark)
    for alt in (False, True):
        args = Namespace(flownet="RAFT", alternate_corr=alt)
        net = fetch_model(args, synthetic_seed=2).to(DEV)
        args.mixed_precision = False
        sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
        truth = oracle_run(sd, "cpu", torch.float64)
        cpu32 = (t(z["flow"]), t(z["g1"]), t(z["g2"]))
        A = product_run(net, args)
        B = oracle_run(sd, DEV, torch.float32)
        C = oracle_run(sd, DEV, torch.float64)
        print(f"== alternate_corr={alt}")
        for name, i in (("flow", 0), ("g1", 1), ("g2", 2)):
            print(f"{name:5s} max|truth| {float(truth[i].abs().max()):.3e}  cpu32 {rel(cpu32[i], truth[i]):.2e}  "
                  f"A(product) {rel(A[i], truth[i]):.2e}  B(torch fp32 gpu) {rel(B[i], truth[i]):.2e}  "
                  f"C(torch f64 gpu) {rel(C[i], truth[i]):.2e}  A vs B {rel(A[i], B[i]):.2e}")


if __name__ == "__main__":
    main()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from argparse import Namespace
import torch, torch.nn.functional as F
from conftest import load_golden, t
from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
from understanding_flow_robustness_amd.flownets import raft_corr

DEV = "cuda:0"
z = load_golden("raft_128x192")
args = Namespace(flownet="RAFT")
net = fetch_model(args, synthetic_seed=2).to(DEV)
args.mixed_precision = False

def torch_lookup(self, coords):
    r = self.radius
    coords = coords.permute(0, 2, 3, 1)
    B, H, W, _ = coords.shape
    out = []
    for i, corr in enumerate(self.corr_pyramid):
        d = torch.linspace(-r, r, 2 * r + 1, device=coords.device)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
        cl = coords.reshape(B * H * W, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        hh, ww = corr.shape[-2:]
        g = torch.cat([2 * cl[..., 0:1] / (ww - 1) - 1, 2 * cl[..., 1:2] / (hh - 1) - 1], -1)
        out.append(F.grid_sample(corr, g, align_corners=True).view(B, H, W, -1))
    return torch.cat(out, -1).permute(0, 3, 1, 2).contiguous().float()

def run(dtype=torch.float32):
    n = net.to(dtype)
    x1, x2 = t(z["x1"], DEV).to(dtype).requires_grad_(True), t(z["x2"], DEV).to(dtype).requires_grad_(True)
    flow = predict_flow(n, None, x1, x2, args)
    loss = (1 - F.cosine_similarity(flow, t(z["target"], DEV).to(dtype))).mean()
    g1, = torch.autograd.grad(loss, x1)
    return flow.detach().double().cpu(), g1.double().cpu()

f_hip, g_hip = run()
orig = raft_corr.CorrBlock.__call__
raft_corr.CorrBlock.__call__ = torch_lookup
f_t, g_t = run()
f64, g64 = f_t, g_t
raft_corr.CorrBlock.__call__ = orig
ref_f, ref_g = t(z["flow"]).double(), t(z["g1"]).double()
rel = lambda a, b: float((a - b).abs().max() / b.abs().max())
print("flow: hip vs ref %.2e | torch-lookup vs ref %.2e | hip vs torch-lookup %.2e | fp64 vs ref %.2e | hip vs fp64 %.2e" % (rel(f_hip, ref_f), rel(f_t, ref_f), rel(f_hip, f_t), rel(f64, ref_f), rel(f_hip, f64)))
print("grad: hip vs ref %.2e | torch-lookup vs ref %.2e | hip vs torch-lookup %.2e | fp64 vs ref %.2e | hip vs fp64 %.2e" % (rel(g_hip, ref_g), rel(g_t, ref_g), rel(g_hip, g_t), rel(g64, ref_g), rel(g_hip, g64)))

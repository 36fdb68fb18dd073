import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from argparse import Namespace
import torch, torch.nn.functional as F
from conftest import load_golden, t
from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
from understanding_flow_robustness_amd.flownets import raft_corr

z = load_golden("raft_128x192")

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
raft_corr.CorrBlock.__call__ = torch_lookup

class IN64(torch.nn.Module):
    def forward(self, x):
        xd = x.double()
        m = xd.mean((2, 3), keepdim=True)
        v = xd.var((2, 3), unbiased=False, keepdim=True)
        return ((xd - m) / torch.sqrt(v + 1e-5)).float()

class IN32(torch.nn.Module):
    def forward(self, x):
        m = x.mean((2, 3), keepdim=True)
        v = x.var((2, 3), unbiased=False, keepdim=True)
        return (x - m) / torch.sqrt(v + 1e-5)

def swap(mod, cls):
    for name, child in mod.named_children():
        if isinstance(child, torch.nn.InstanceNorm2d):
            setattr(mod, name, cls())
        elif isinstance(child, torch.nn.Sequential) and name == "downsample":
            for i, c in enumerate(child):
                if isinstance(c, torch.nn.InstanceNorm2d):
                    child[i] = cls()
        else:
            swap(child, cls)

MODE = None
def run(dev):
    args = Namespace(flownet="RAFT")
    net = fetch_model(args, synthetic_seed=2).to(dev)
    if MODE is not None and dev != "cpu":
        swap(net.fnet, MODE)
    saved = {}
    def hook(name):
        def f(mod, inp, out):
            o = out if torch.is_tensor(out) else out[0]
            o.register_hook(lambda g: saved.__setitem__(name, g.detach().double().cpu()))
        return f
    net.fnet.register_forward_hook(hook("fnet_out"))
    net.cnet.register_forward_hook(hook("cnet_out"))
    net.fnet.layer1.register_forward_hook(hook("fnet_layer1"))
    net.fnet.layer3.register_forward_hook(hook("fnet_layer3"))
    net.fnet.relu1.register_forward_hook(hook("fnet_stem"))
    net.cnet.layer1.register_forward_hook(hook("cnet_layer1"))
    net.update_block.gru.register_forward_hook(hook("gru_last"))
    x1, x2 = t(z["x1"], dev).requires_grad_(True), t(z["x2"], dev).requires_grad_(True)
    _, flow = net(x1 * 255.0, x2 * 255.0, test_mode=True)
    loss = (1 - F.cosine_similarity(flow, t(z["target"], dev))).mean()
    g1, g2 = torch.autograd.grad(loss, (x1, x2))
    saved["image1"], saved["image2"] = g1.double().cpu(), g2.double().cpu()
    return saved
a = run("cpu")
for mode in (None, IN32, IN64):
    MODE = mode
    torch.backends.cudnn.benchmark = False
    b = run("cuda:0")
    print("== GPU instance norm:", "native" if mode is None else mode.__name__)
    for k in a:
        print(f"{k:14s} rel err {float((a[k]-b[k]).abs().max()/a[k].abs().max()):.2e}  (max {float(a[k].abs().max()):.2e})")
print("cpu image1 vs golden", float((a["image1"] - t(z["g1"]).double()).abs().max() / t(z["g1"]).abs().max()))

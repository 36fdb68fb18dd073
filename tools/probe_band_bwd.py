"""Probe: MIOpen backward-data time of FlowNetC head convolutions at full width vs a column band."""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda:0"
B = 8
layers = [  # name, cin, cout, k, s, p, H_in, W_in, band fraction of input columns that need gradients
    ("conv3_1", 473, 256, 3, 1, 1, 48, 160, 58 / 160), ("conv4", 256, 512, 3, 2, 1, 48, 160, 58 / 160),
    ("conv4_1", 512, 512, 3, 1, 1, 24, 80, 30 / 80), ("conv5", 512, 512, 3, 2, 1, 24, 80, 32 / 80),
    ("conv5_1", 512, 512, 3, 1, 1, 12, 40, 17 / 40), ("conv6", 512, 1024, 3, 2, 1, 12, 40, 19 / 40),
    ("conv6_1", 1024, 1024, 3, 1, 1, 6, 20, 11 / 20)]
def t_bwd(cin, cout, k, s, p, H, W):
    x = torch.randn(B, cin, H, W, device=dev, requires_grad=True)
    w = torch.randn(cout, cin, k, k, device=dev)
    y = F.conv2d(x, w, None, s, p)
    g = torch.randn_like(y)
    for _ in range(3):
        torch.autograd.grad(y, x, g, retain_graph=True)
    torch.cuda.synchronize()
    s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record()
    for _ in range(10):
        torch.autograd.grad(y, x, g, retain_graph=True)
    e0.record(); e0.synchronize()
    return s0.elapsed_time(e0) / 10
tot_f = tot_b = 0
for name, cin, cout, k, s, p, H, W, frac in layers:
    full = t_bwd(cin, cout, k, s, p, H, W)
    wb = min(W, int(W * frac) + 4 * s)
    wb += wb % 2
    band = t_bwd(cin, cout, k, s, p, H, wb)
    tot_f += full; tot_b += band
    print(f"{name:8s} full W={W:4d}: {full*1e3:7.1f} us   band W={wb:4d}: {band*1e3:7.1f} us", flush=True)
print(f"sum full {tot_f:.3f} ms, band {tot_b:.3f} ms")

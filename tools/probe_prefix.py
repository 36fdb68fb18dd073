"""Probe: MIOpen time of FlowNetC's conv1-3 at full frame (once per attack() call) and on the 128x128 window."""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda:0"


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / 10 * 1e3


for tag, H, W in (("full frame 384x1280", 384, 1280), ("window 128x128", 128, 128)):
    h, w = H, W
    for name, cin, cout, k in (("conv1", 3, 64, 7), ("conv2", 64, 128, 5), ("conv3", 128, 256, 5)):
        x = torch.randn(16, cin, h, w, device=dev, requires_grad=True)
        wt = torch.randn(cout, cin, k, k, device=dev)
        y = F.conv2d(x, wt, None, 2, (k - 1) // 2)
        g = torch.randn_like(y)
        tf = timeit(lambda: F.conv2d(x, wt, None, 2, (k - 1) // 2))
        tb = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
        gf = 2 * cin * cout * k * k * y.shape[2] * y.shape[3] * 16 / 1e9
        print(f"{tag:20s} {name}: fwd {tf:8.1f} us ({gf / tf * 1e3:6.1f} TF/s)   bwd-data {tb:8.1f} us", flush=True)
        h, w = h // 2, w // 2

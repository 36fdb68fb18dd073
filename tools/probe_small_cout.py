"""Probe: MIOpen time of FlowNetC's 2-channel layers (predict_flow*, upsampled_flow*) at batch 8, 384x1280."""
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev, B = "cuda:0", 8


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


tot_f = tot_b = 0.0
for name, cin, h, w in (("predict_flow6", 1024, 6, 20), ("predict_flow5", 1026, 12, 40), ("predict_flow4", 770, 24, 80),
                        ("predict_flow3", 386, 48, 160), ("predict_flow2", 194, 96, 320)):
    x = torch.randn(B, cin, h, w, device=dev, requires_grad=True)
    wgt, b = torch.randn(2, cin, 3, 3, device=dev), torch.randn(2, device=dev)
    y = F.conv2d(x, wgt, b, 1, 1)
    g = torch.randn_like(y)
    tf = timeit(lambda: F.conv2d(x, wgt, b, 1, 1))
    tb = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    tot_f += tf; tot_b += tb
    print(f"{name:16s} fwd {tf:7.1f} us   bwd-data {tb:7.1f} us   (input {B*cin*h*w*4/1e6:6.1f} MB)", flush=True)
for name, h, w in (("up6to5", 6, 20), ("up5to4", 12, 40), ("up4to3", 24, 80), ("up3to2", 48, 160)):
    x = torch.randn(B, 2, h, w, device=dev, requires_grad=True)
    wgt, b = torch.randn(2, 2, 4, 4, device=dev), torch.randn(2, device=dev)
    y = F.conv_transpose2d(x, wgt, b, 2, 1)
    g = torch.randn_like(y)
    tf = timeit(lambda: F.conv_transpose2d(x, wgt, b, 2, 1))
    tb = timeit(lambda: torch.autograd.grad(y, x, g, retain_graph=True))
    tot_f += tf; tot_b += tb
    print(f"{name:16s} fwd {tf:7.1f} us   bwd-data {tb:7.1f} us", flush=True)
print(f"total fwd {tot_f/1e3:.3f} ms, bwd {tot_b/1e3:.3f} ms per iteration")

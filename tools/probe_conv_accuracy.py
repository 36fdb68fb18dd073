"""How accurate is the fp32 convolution the attack runs on today, and how accurate would the split-precision GEMM
be?  A 256 -> 256 3x3 layer at FlowNetC's 1/8 resolution: float64 on the CPU is the truth; MIOpen's fp32 kernel
(whatever its find step picks: Winograd or implicit GEMM), torch's fp32 matmul on the im2col matrix and
csrc/split_gemm.hip with 6 / 3 / 1 products are measured against it.  Then timings at 8 pairs."""
import json
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from understanding_flow_robustness_amd.split_gemm import gemm_split_nt, split_bf16x3  # noqa: E402

DEV = "cuda:0"
torch.backends.cudnn.benchmark = True
g = torch.Generator().manual_seed(0)
C, K, H, W = 256, 256, 48, 160
x = torch.randn(1, C, H, W, generator=g)
w = torch.randn(K, C, 3, 3, generator=g) * (2.0 / (9 * C)) ** 0.5
ref = F.conv2d(x.double(), w.double(), padding=1)
scale = float(ref.abs().max())


def err(y):
    return float((y.detach().cpu().double() - ref).abs().max()) / scale


out = {"layer": f"{C}->{K} 3x3 at {H}x{W}", "cpu_fp32": err(F.conv2d(x, w, padding=1))}
xd, wd = x.to(DEV), w.to(DEV)
out["miopen_fp32"] = err(F.conv2d(xd, wd, padding=1))
cols = F.unfold(xd, 3, padding=1)[0].t().contiguous()            # [H*W, C*9]
wm = wd.reshape(K, C * 9).contiguous()
out["torch_matmul_fp32"] = err((cols @ wm.t()).t().reshape(1, K, H, W))
ap, bp = split_bf16x3(cols), split_bf16x3(wm)
for products in (6, 3, 1):
    out[f"split_{products}"] = err(gemm_split_nt(ap, bp, products).t().reshape(1, K, H, W))
# the adjoint is what MIOpen computes with split-K / Winograd variants too
gy = torch.randn(1, K, H, W, generator=g)
gref = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
gx = torch.ops.aten.convolution_backward(gy.to(DEV), xd, wd, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                         [True, False, False])[0]
out["miopen_fp32_bwd_data"] = float((gx.cpu().double() - gref).abs().max()) / float(gref.abs().max())
print(json.dumps(out), flush=True)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


B = 8
xb = torch.randn(B, C, H, W, device=DEV)
flop = 2.0 * B * H * W * K * C * 9
t = {"pairs": B, "gflop": round(flop * 1e-9, 1)}
t["miopen_ms"] = round(timed(lambda: F.conv2d(xb, wd, padding=1)), 4)
colsb = torch.randn(B * H * W, C * 9, device=DEV)
apb = split_bf16x3(colsb)
for products in (6, 3):
    t[f"split_{products}_gemm_ms"] = round(timed(lambda: gemm_split_nt(apb, bp, products)), 4)
t["miopen_eff_tflops"] = round(flop / t["miopen_ms"] * 1e-9, 1)
print(json.dumps(t), flush=True)

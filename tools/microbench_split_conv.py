"""FlowNetC's 3x3 stride-1 head layers at 8 pairs of 384x1280: MIOpen fp32 (find-selected) against the
split-precision implicit GEMM (csrc/split_gemm.hip), the NCHW -> NHWC-planes pass timed separately."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, ".")
from understanding_flow_robustness_amd.split_gemm import (chunk_major, conv3x3_split, conv3x3_weight_planes,  # noqa: E402
                                                          nchw_to_nhwc_split3)

DEV = "cuda:0"
torch.backends.cudnn.benchmark = True


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


for name, C, N, H, W in (("conv3_1", 473, 256, 48, 160), ("conv4_1", 512, 512, 24, 80), ("conv5_1", 512, 512, 12, 40),
                         ("conv6_1", 1024, 1024, 6, 20)):
    B = 8
    x = torch.randn(B, C, H, W, device=DEV)
    w = torch.randn(N, C, 3, 3, device=DEV) * (2.0 / (9 * C)) ** 0.5
    flop = 2.0 * B * H * W * N * C * 9
    row = dict(layer=name, gflop=round(flop * 1e-9, 1))
    skip_miopen = bool(os.environ.get("UFR_SKIP_MIOPEN"))           # its find step dominates the script's run time
    if not skip_miopen:
        ms = timed(lambda: F.conv2d(x, w, padding=1))
        row["miopen_ms"], row["miopen_tflops"] = round(ms, 4), round(flop / ms * 1e-9, 1)
    xp, wp = nchw_to_nhwc_split3(x), conv3x3_weight_planes(w)
    row["to_planes_ms"] = round(timed(lambda: nchw_to_nhwc_split3(x)), 4)
    for products in (6, 3):
        ms = timed(lambda: conv3x3_split(xp, wp, B, H, W, products))
        row[f"split{products}_ms"], row[f"split{products}_tflops"] = round(ms, 4), round(flop / ms * 1e-9, 1)
    xc, wc = chunk_major(xp), chunk_major(wp)
    for products in (6, 3):
        ms = timed(lambda: conv3x3_split(xc, wc, B, H, W, products, chunked=True))
        row[f"chunked{products}_ms"], row[f"chunked{products}_tflops"] = round(ms, 4), round(flop / ms * 1e-9, 1)
    if os.environ.get("UFR_EXPERIMENTAL") and N % 256 == 0:           # csrc/split_conv_wide.hip
        for products in (6, 3):
            ms = timed(lambda: conv3x3_split(xc, wc, B, H, W, products, chunked=True, wide=True))
            row[f"wide{products}_ms"], row[f"wide{products}_tflops"] = round(ms, 4), round(flop / ms * 1e-9, 1)
    if not skip_miopen:
        ref = F.conv2d(x[:1], w, padding=1)
        y = conv3x3_split(xp, wp, B, H, W, 6)[: H * W, :N].reshape(1, H, W, N).permute(0, 3, 1, 2)
        row["max_diff_vs_miopen"] = float((y - ref).abs().max() / ref.abs().max())
    print(json.dumps(row), flush=True)

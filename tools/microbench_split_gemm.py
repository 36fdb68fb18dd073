"""Times csrc/split_gemm.hip against torch's float32 matmul (hipBLASLt/rocBLAS) on GEMM shapes of FlowNetC's head
(M = pixels of 8 pairs, N = output channels, K = 9 * input channels).  One JSON line per shape."""
import json
import sys

import torch

sys.path.insert(0, ".")
from understanding_flow_robustness_amd.split_gemm import gemm_split_nt, split_bf16x3  # noqa: E402

DEV = "cuda:0"


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


for name, M, N, K in (("conv3_1", 61440, 256, 4256), ("conv4_1", 15360, 512, 4608), ("conv6_1", 1024, 1024, 9216),
                      ("square", 4096, 4096, 4096), ("big", 8192, 8192, 8192)):
    a, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV)
    ap, bp = split_bf16x3(a), split_bf16x3(b)
    flop = 2.0 * M * N * K
    row = dict(shape=name, M=M, N=N, K=K)
    row["split_ms"] = round(timed(lambda: split_bf16x3(a)), 4)
    for products in (6, 3, 1):
        ms = timed(lambda: gemm_split_nt(ap, bp, products))
        row[f"p{products}_ms"] = round(ms, 4)
        row[f"p{products}_eff_tflops"] = round(flop / ms * 1e-9, 1)          # float32-equivalent rate
        row[f"p{products}_mfma_tflops"] = round(products * flop / ms * 1e-9, 1)
    bt = b.t().contiguous()
    ms = timed(lambda: a @ bt)
    row["torch_fp32_ms"], row["torch_fp32_tflops"] = round(ms, 4), round(flop / ms * 1e-9, 1)
    print(json.dumps(row), flush=True)

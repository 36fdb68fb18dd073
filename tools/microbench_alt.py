"""alt_cuda_corr backward micro-benchmark (RAFT shapes: 48x160 pixels, 256 channels, radius 4, 4 levels).

    UFR_ALTCORR_BWD_VARIANT=0|1 python tools/microbench_alt.py
Prints the average duration of one backward call per pyramid level for a smooth flow (identity + 0.3 px
noise), a moderately rough one (3 px) and a wild one (40 px)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from understanding_flow_robustness_amd import alt_cuda_corr
    dev = "cuda:0"
    g = torch.Generator().manual_seed(0)
    B, H, W, C, r = 1, 48, 160, 256, 4
    f1 = torch.randn(B, H, W, C, generator=g).to(dev)
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    base = torch.stack([xs, ys], -1)
    print("variant", os.environ.get("UFR_ALTCORR_BWD_VARIANT", "1"))
    for name, spread in (("smooth", 0.3), ("rough", 3.0), ("wild", 40.0)):
        coords0 = (base + spread * torch.randn(B, 1, H, W, 2, generator=g)).to(dev)
        for lvl in range(4):
            f2 = torch.randn(B, H >> lvl, W >> lvl, C, generator=g).to(dev)
            coords = (coords0 / 2 ** lvl).contiguous()
            (o,) = alt_cuda_corr.forward(f1, f2, coords, r)
            go = torch.randn_like(o)
            for _ in range(3):
                alt_cuda_corr.backward(f1, f2, coords, go, r)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(20):
                alt_cuda_corr.backward(f1, f2, coords, go, r)
            e.record()
            e.synchronize()
            t_b = s.elapsed_time(e) / 20 * 1e3
            s.record()
            for _ in range(20):
                alt_cuda_corr.forward(f1, f2, coords, r)
            e.record()
            e.synchronize()
            print(f"{name:7s} level {lvl}: {t_b:8.1f} us / backward call, {s.elapsed_time(e) / 20 * 1e3:7.1f} us / forward call",
                  flush=True)


if __name__ == "__main__":
    main()

"""Timing experiments for csrc/correlation_planes.hip: one launch at FlowNetC's 1/8 grid for several batch sizes (B = 1 is
a single round of workgroups: the latency of one workgroup), under the UFR_CORR_DEBUG switches (1 no stores, 2 no loads,
4 no MFMA) -- the output checksum shows whether a switch took effect."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from understanding_flow_robustness_amd import _lib as L
from understanding_flow_robustness_amd import igemm as ig

DEV = "cuda:0"


def main():
    lib = L.lib()
    H, W = 48, 160
    for B in (1, 2, 4, 8):
        g = torch.Generator().manual_seed(0)
        f1 = ig.Planes(B, H, W, 8, DEV).load_nchw(torch.randn(B, 256, H, W, generator=g).to(DEV), 0)
        f2 = ig.Planes(B, H, W, 8, DEV).load_nchw(torch.randn(B, 256, H, W, generator=g).to(DEV), 0)
        out = ig.Planes(B, H, W, 15, DEV)
        fn = lambda: L.check(lib.ufr_corr_forward_planes(L.ptr(f1.t), L.ptr(f2.t), f1.plane_stride, L.ptr(out.t), out.plane_stride, 1,
                                                         B, 256, H, W, 21, 2, 1.0 / 256.0, 0.1, L.stream()))
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fn()
        e.record()
        e.synchronize()
        ms = s.elapsed_time(e) / 20
        print(json.dumps(dict(debug=os.environ.get("UFR_CORR_DEBUG", "0"), B=B, workgroups=B * H * 2, ms=round(ms, 4),
                              checksum=float(out.t.float().abs().sum()))), flush=True)


if __name__ == "__main__":
    main()

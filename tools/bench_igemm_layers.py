"""Per-layer timings of csrc/igemm.hip on FlowNetC's head at the benchmark size (8 pairs, 384x1280): forward and data
gradient of every `conv` / `deconv` block (models/FlowNetC.py:22-50), HIP events on the launch stream.
One JSON line per launch: ms, fp32-equivalent TFLOP/s (2*M*N*K / time), fraction of the 417 TFLOP/s six-product ceiling."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from understanding_flow_robustness_amd import igemm as ig

DEV = "cuda:0"
B, H8, W8 = 8, 48, 160
CEIL = 2500.0 / 6


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    only = set(a for a in sys.argv[1:] if not a.startswith("--"))
    variants = (4, 2, 4, 2) if "--m64" in sys.argv else ((5, 2, 5, 2) if "--pipe" in sys.argv else ((6, 5, 6, 5) if "--pp" in sys.argv else (6,)))
    layers = [  # name, kind, Cin, Cout, k, stride, input grid (of the forward)
        ("conv3_1", "conv", 473, 256, 3, 1, (H8, W8)), ("conv4", "conv", 256, 512, 3, 2, (H8, W8)),
        ("conv4_1", "conv", 512, 512, 3, 1, (H8 // 2, W8 // 2)), ("conv5", "conv", 512, 512, 3, 2, (H8 // 2, W8 // 2)),
        ("conv5_1", "conv", 512, 512, 3, 1, (H8 // 4, W8 // 4)), ("conv6", "conv", 512, 1024, 3, 2, (H8 // 4, W8 // 4)),
        ("conv6_1", "conv", 1024, 1024, 3, 1, (H8 // 8, W8 // 8)),
        ("deconv5", "deconv", 1024, 512, 4, 2, (H8 // 8, W8 // 8)), ("deconv4", "deconv", 1026, 256, 4, 2, (H8 // 4, W8 // 4)),
        ("deconv3", "deconv", 770, 128, 4, 2, (H8 // 2, W8 // 2)), ("deconv2", "deconv", 386, 64, 4, 2, (H8, W8)),
    ]
    g = torch.Generator().manual_seed(0)
    for name, kind, cin, cout, k, s, (hi, wi) in layers:
        if only and name not in only:
            continue
        p = (k - 1) // 2 if kind == "conv" else 1
        if kind == "conv":
            w = (torch.randn(cout, cin, k, k, generator=g) * 0.03).to(DEV)
            ho, wo = (hi + 2 * p - k) // s + 1, (wi + 2 * p - k) // s + 1
            fwd_w, bwd_w = ig.conv_forward_weights(w, s, p), ig.conv_backward_weights(w, s, p)
            fwd_rows, bwd_rows = (ho, wo), (ho, wo)
        else:
            w = (torch.randn(cin, cout, k, k, generator=g) * 0.03).to(DEV)
            ho, wo = 2 * hi, 2 * wi
            fwd_w, bwd_w = ig.deconv_forward_weights(w, p), ig.deconv_backward_weights(w, p)
            fwd_rows, bwd_rows = (hi, wi), (hi, wi)
        bias = torch.randn(cout, generator=g).to(DEV)
        x = ig.Planes(B, hi, wi, ig.pad32(cin) // 32, DEV).load_nchw(torch.randn(B, cin, hi, wi, generator=g).to(DEV))
        y = ig.Planes(B, ho, wo, ig.pad32(cout) // 32, DEV)
        gy = ig.Planes(B, ho, wo, ig.pad32(cout) // 32, DEV).load_nchw(torch.randn(B, cout, ho, wo, generator=g).to(DEV))
        gx = ig.Planes(B, hi, wi, ig.pad32(cin) // 32, DEV)
        flop = 2.0 * B * ho * wo * cout * cin * k * k / (s * s if kind == "deconv" else 1)
        for tag, wimg, src, rows, out_hw, dst, kw in (("fwd", fwd_w, x, fwd_rows, (ho, wo), y, dict(bias=bias)),
                                                     ("bwd", bwd_w, gy, bwd_rows, (hi, wi), gx, dict(mask=x))):
            M = B * rows[0] * rows[1]
            ktiles = max(len(t) for _, _, t in wimg.phases) * wimg.KC
            S0 = ig.splitk_for(M, wimg.Npad, ktiles, len(wimg.phases))
            cands = sorted({1, S0}) if "--m64" not in sys.argv else sorted({1, max(1, S0 // 2), S0})
            if "--pp" in sys.argv or "--pp3" in sys.argv:         # ping-pong: one 256-row workgroup per CU
                cands = sorted({1, ig.splitk_for(M, wimg.Npad, ktiles, len(wimg.phases), target=512),
                                ig.splitk_for(M, wimg.Npad, ktiles, len(wimg.phases), target=256, bm=256)})
            if "--pipe" in sys.argv:       # two workgroups per CU: 512 slots
                cands = sorted({1, S0, ig.splitk_for(M, wimg.Npad, ktiles, len(wimg.phases), target=512)})
            for S, variant in [(S, v) for S in cands for v in variants]:
                ws = torch.empty(len(wimg.phases) * S * M * wimg.Npad, device=DEV) if S > 1 else None
                launch = ig.make_launch(wimg, src, 0, rows, out_hw, out_planes=dst, splitk=S, ws=ws, variant=variant, **kw)
                ms = timed(launch)
                tf = flop / ms / 1e9
                print(json.dumps(dict(layer=name, dir=tag, variant=variant, splitk=S, M=M, Npad=wimg.Npad, ktiles=ktiles, phases=len(wimg.phases),
                                      gflop=round(flop / 1e9, 1), ms=round(ms, 4), tflops=round(tf, 1), frac_of_417=round(tf / CEIL, 3))),
                      flush=True)


if __name__ == "__main__":
    main()

"""Per-kernel timing of the gfx950 operators at BASELINE shapes (HIP events on the launch stream)."""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from understanding_flow_robustness_amd import spatial_correlation_sampler_backend as be


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    dev = "cuda:0"
    res = {}
    for B in (1, 8):
        a = torch.randn(B, 256, 48, 160, device=dev)
        b = torch.randn(B, 256, 48, 160, device=dev)
        prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
        out = be.forward(a, b, *prm)
        go = torch.randn_like(out)
        tf = timeit(lambda: be.forward(a, b, *prm))
        tb = timeit(lambda: be.backward(a, b, go, *prm))
        res[f"flownetc_corr_B{B}"] = dict(fwd_ms=tf, bwd_ms=tb, fwd_tflops=1.734e-3 * B / tf, bwd_tflops=3.468e-3 * B / tb)
    for (C, H, W) in ((196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)):
        B = 8
        a = torch.randn(B, C, H, W, device=dev); b = torch.randn(B, C, H, W, device=dev)
        prm = (1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
        out = be.forward(a, b, *prm); go = torch.randn_like(out)
        res[f"pwc_corr_C{C}_{H}x{W}_B8"] = dict(fwd_ms=timeit(lambda: be.forward(a, b, *prm)),
                                               bwd_ms=timeit(lambda: be.backward(a, b, go, *prm)))
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()

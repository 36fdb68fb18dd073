"""Where predict_flow's forward kernel spends its time on the small grids: the same launch at 1 .. 32 chunks and 1 .. 8 frames.

    python tools/bench_pf.py
One JSON line per (grid, frames, chunks): microseconds per launch (HIP events around 200 back-to-back launches)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from understanding_flow_robustness_amd import _lib as L  # noqa: E402
from understanding_flow_robustness_amd import igemm as ig  # noqa: E402
from understanding_flow_robustness_amd.flownetc_engine import _pack_flow_head_mfma  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters=200):
    for _ in range(5):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    lib = L.lib()
    for (H, W), cmax in (((6, 20), 32), ((12, 40), 33), ((24, 80), 25), ((48, 160), 13)):
        for B in (1, 8):
            planes = ig.Planes(B, H, W, cmax, DEV)
            planes.t.normal_()
            w = torch.randn(2, cmax * 32, 3, 3, device=DEV)
            wm = _pack_flow_head_mfma(w)
            bias = torch.zeros(2, device=DEV)
            out = torch.zeros(B, 2, H, W, device=DEV)
            # never more chunks than the buffer and the packed weights hold: round 4's form of this loop launched 16 chunks on
            # the 13-chunk buffer of the 48 x 160 grid and the kernel read past its end -- a GPU memory-access fault at 8 frames
            # (DESIGN.md 6.5); since ABI 7 the C entry refuses such a range as well
            for chunks in sorted(c for c in {1, 4, 8, 16, cmax} if c <= cmax):
                fn = lambda: L.check(lib.ufr_flow_head_planes_forward_mfma(L.ptr(planes.t), planes.plane_stride, 0, chunks, L.ptr(wm),
                                                                           wm.shape[0], L.ptr(bias), L.ptr(out), B, H, W, L.stream()), "pf")
                print(json.dumps(dict(grid=[H, W], frames=B, chunks=chunks, us=round(timed(fn), 2))), flush=True)
    # the floor: an empty-ish kernel of torch
    x = torch.zeros(64, device=DEV)
    print(json.dumps(dict(kernel="torch add_ on 64 floats", us=round(timed(lambda: x.add_(1.0)), 2))), flush=True)


if __name__ == "__main__":
    main()

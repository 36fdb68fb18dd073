"""Bounding boxes of the alt_corr lookup's windows in the C3 bench configuration (RAFT, synthetic weights, 384x1280, one pair):
for every refinement iteration and pyramid level, the mean / max box (rows x columns) of a 16 x 1 pixel tile (round 3's kernel)
and of an 8 x 16 tile (round 6's), from the coordinates the engine actually looked up."""
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow  # noqa: E402

DEV = "cuda:0"
args = Namespace(flownet="RAFT", alternate_corr=True)
net = fetch_model(args, synthetic_seed=2).to(DEV).eval().requires_grad_(False)
args.mixed_precision = False
g = torch.Generator().manual_seed(0)
tgt, ref = torch.rand(1, 3, 384, 1280, generator=g).to(DEV), torch.rand(1, 3, 384, 1280, generator=g).to(DEV)
with torch.no_grad():
    predict_flow(net, None, tgt, ref, args)
eng = next(iter(net.__dict__["_ufr_head_engines"].values()))
for it in (0, 1, 5, 11):
    c = eng.coords_it[it][0].cpu()                     # [2, 48, 160]
    flow = c - eng.coords0[0].cpu()
    line = f"iteration {it}: |flow| mean {float(flow.abs().mean()):.2f} max {float(flow.abs().max()):.1f}; "
    for l in range(4):
        x, y = torch.floor(c[0] / 2 ** l), torch.floor(c[1] / 2 ** l)
        out = []
        for th, tw in ((1, 16), (8, 16)):
            xs = x.unfold(0, th, th).unfold(1, tw, tw)
            ys = y.unfold(0, th, th).unfold(1, tw, tw)
            w = (xs.amax((-1, -2)) - xs.amin((-1, -2)) + 10)
            h = (ys.amax((-1, -2)) - ys.amin((-1, -2)) + 10)
            out.append(f"{th}x{tw}: {float(h.mean()):.0f}x{float(w.mean()):.0f} (max {int(h.max())}x{int(w.max())})")
        line += f"L{l} " + ", ".join(out) + "; "
    print(line)

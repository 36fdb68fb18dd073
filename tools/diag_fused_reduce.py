"""Which prepared launch of an engine differs between the fused split-K reduction (`tickets`) and the slab kernel + reduce launch?
    python tools/diag_fused_reduce.py            (RAFT feature encoder, 2 frames of 64 x 128)
Every plan of the engine is built twice over the same operands (random planes), run from the same initial output, compared bit for bit."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from argparse import Namespace  # noqa: E402

from understanding_flow_robustness_amd import igemm as ig  # noqa: E402
from understanding_flow_robustness_amd.flownets.utils_model import fetch_model  # noqa: E402
from understanding_flow_robustness_amd.raft_encoder_engine import RaftEncoderEngine  # noqa: E402

DEV = "cuda:0"
n, H, W = (int(v) for v in (sys.argv[1:4] + ["2", "64", "128"][len(sys.argv) - 1:]))
net = fetch_model(Namespace(flownet="RAFT"), synthetic_seed=2).to(DEV).eval().requires_grad_(False)
eng = RaftEncoderEngine(net.fnet, n, H, W, DEV)
g = torch.Generator(device=DEV).manual_seed(0)
bad = 0
for i, (holder, wi, xin, rows, out_hw, kw) in enumerate(eng._plans):
    d = holder.launch.desc
    if d.splitk <= 1:
        continue
    xin.t.copy_((torch.randn(xin.t.shape, device=DEV, generator=g) * 0.5).to(xin.t.dtype))
    out = kw.get("out_f32") if kw.get("out_f32") is not None else kw.get("out_planes")
    init = (torch.randn(out.t.shape, device=DEV, generator=g)).to(out.t.dtype)
    res = []
    for fuse in (False, True):
        ws = torch.full((len(wi.phases) * d.splitk * n * rows[0] * rows[1] * wi.Npad,), float("nan"), device=DEV)
        launch = ig.make_launch(wi, xin, 0, rows, out_hw, splitk=d.splitk, ws=ws, fuse_reduce=fuse, **kw)
        out.t.copy_(init)
        launch()
        torch.cuda.synchronize()
        res.append(out.t.float().clone())
    same = torch.equal(res[0], res[1])
    diff = (res[0] - res[1]).abs()
    nbad = int((diff > 0).sum()) + int((torch.isnan(res[1]) != torch.isnan(res[0])).sum())
    bad += not same
    taps = "-".join(str(len(t)) for _, _, t in wi.phases)
    print(f"plan {i:2d}: M={n * rows[0] * rows[1]:6d} rows={rows} out={out_hw} Npad={wi.Npad} N={wi.N} KC={wi.KC} taps={taps} splitk={d.splitk} variant={d.variant} "
          f"add={'inplace' if kw.get('add') is out else kw.get('add') is not None} -> {'same' if same else f'DIFFERENT in {nbad} of {diff.numel()} (max {float(diff.nan_to_num(9e9).max()):.3e})'}",
          flush=True)
    if not same:
        idx = (diff > 0).nonzero()[:6].tolist()
        print("   first differing (chunk, pixel, lane):", idx)
print("differing launches:", bad)

"""How far do two float32 summation orders of the SAME full-frame attack step drift apart over 1, 2, 3 iterations?
The control for tests/test_cone_gpu.py::test_windowed_step_random_placements: full-frame step with the single-stage igemm
(UFR_IGEMM_PIPE=0) against the full-frame step with the pipelined one (different split-K sizes -> another rounding), next
to the windowed step against the full-frame one (same kernels).  Prints, per attack() call, the share of patch pixels that
differ by more than 1e-4 of the update and the worst difference as a fraction of the update."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from argparse import Namespace

import torch

import test_cone_gpu as T

DEV = "cuda:0"


def fresh(net):
    for k in [k for k in net.__dict__ if k.startswith("_ufr")]:
        net.__dict__.pop(k)


def stats(pf, pc, p0, sel):
    upd = float(((pf - p0) * sel).abs().max())
    err = ((pf - pc) * sel).abs()
    off = float((err > 1e-4 * upd + 1e-6).sum()) / max(float((sel != 0).sum()), 1.0)
    return f"{off:.2%} off, worst {float(err.max()) / upd:.2e}"


def main():
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    net = fetch_model(Namespace(flownet="FlowNetC"), synthetic_seed=0).to(DEV)
    B, H, W = 8, 384, 1280
    g = torch.Generator().manual_seed(1234)
    yy, xx = torch.meshgrid(torch.arange(51, device=DEV), torch.arange(51, device=DEV), indexing="ij")
    disc = (((yy - 25) ** 2 + (xx - 25) ** 2) <= 23 ** 2).float()
    masks = []
    for _ in range(3):
        m = torch.zeros(B, 3, H, W, device=DEV)
        for b in range(B):
            y = int(torch.randint(0, H - 51 + 1, (1,), generator=g))
            x = int(torch.randint(0, W - 51 + 1, (1,), generator=g))
            m[b, :, y:y + 51, x:x + 51] = disc
        masks.append(m)
    lr = T._unclamped_lr(net, masks[0], B, H, W, False)
    for iters in (1, 2, 3):
        runs = {}
        for pipe in ("0", "1"):
            os.environ["UFR_IGEMM_PIPE"] = pipe
            fresh(net)
            _, p0, runs["full", pipe] = T._run_step(net, False, masks, B, H, W, lr, False, iters=iters)
            fresh(net)
            _, _, runs["cone", pipe] = T._run_step(net, True, masks, B, H, W, lr, False, iters=iters)
        for c, mask in enumerate(masks):
            print(f"iters={iters} call={c}: full(single-stage) vs full(pipelined): {stats(runs['full', '0'][c][0], runs['full', '1'][c][0], p0, mask)}"
                  f" | cone vs full, pipelined: {stats(runs['full', '1'][c][0], runs['cone', '1'][c][0], p0, mask)}"
                  f" | cone vs full, single-stage: {stats(runs['full', '0'][c][0], runs['cone', '0'][c][0], p0, mask)}", flush=True)


if __name__ == "__main__":
    main()

"""Diagnostic: windowed step (incremental head forward on / off) against the full-frame step, per pair."""
import os
import sys
from argparse import Namespace

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
DEV = "cuda:0"


def main():
    import test_cone_gpu as T
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    net = fetch_model(Namespace(flownet="FlowNetC"), synthetic_seed=0).to(DEV)
    B, H, W = 8, 384, 1280
    g = torch.Generator().manual_seed(1234)
    yy, xx = torch.meshgrid(torch.arange(51, device=DEV), torch.arange(51, device=DEV), indexing="ij")
    disc = (((yy - 25) ** 2 + (xx - 25) ** 2) <= 23 ** 2).float()
    masks, places = [], []
    for _ in range(3):
        m = torch.zeros(B, 3, H, W, device=DEV)
        pl = []
        for b in range(B):
            y = int(torch.randint(0, H - 51 + 1, (1,), generator=g)); x = int(torch.randint(0, W - 51 + 1, (1,), generator=g))
            m[b, :, y:y + 51, x:x + 51] = disc
            pl.append((y, x))
        masks.append(m); places.append(pl)
    lr = T._unclamped_lr(net, masks[0], B, H, W, False)
    for iters in (2, 3):
        _, p0, full = T._run_step(net, False, masks, B, H, W, lr, False, iters=iters)
        for inc in ("1", "0"):
            os.environ["UFR_INCREMENTAL"] = inc
            _, _, cone = T._run_step(net, True, masks, B, H, W, lr, False, iters=iters)
            for c, ((pf, _, _, _), (pc, _, _, _), mask, pl) in enumerate(zip(full, cone, masks, places)):
                upd = ((pf - p0) * mask).abs().amax(dim=(1, 2, 3))
                err = ((pf - pc) * mask).abs().amax(dim=(1, 2, 3))
                rel = (err / upd.clamp_min(1e-9)).tolist()
                worst = max(range(B), key=lambda b: rel[b])
                print(f"iters {iters} incremental {inc} call {c}: worst rel {rel[worst]:.2e} at pair {worst} place {pl[worst]}; "
                      f"median {sorted(rel)[B // 2]:.2e}", flush=True)


if __name__ == "__main__":
    main()

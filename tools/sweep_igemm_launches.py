"""Sweep the kernel form and the split-K factor of the engine's SMALL igemm launches on the step's own buffers.

    python tools/sweep_igemm_launches.py [window] [s2bwd] [small]

For every chosen launch of the headline step (FlowNetC 384x1280, 8 pairs; buffers hold real activations after two iterations)
the launch is rebuilt with every (variant, split-K) pair and timed with HIP events, the engine's own choice first.  One JSON
line per (launch, variant, splitk): ms, fraction of the six-product ceiling.
  window : conv2 / conv3 forward and the three data gradients of the 128x128 attack window
  s2bwd  : the stride-2 data gradients (conv4 / conv5 / conv6 bwd, full and band)
  small  : the 1/32 .. 1/64 grids (conv5_1, conv6, conv6_1, deconv5 forward and backward)
"""
import json
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from understanding_flow_robustness_amd import igemm as ig  # noqa: E402

DEV = "cuda:0"
CEIL = 2500.0 / 6


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def sweep(label, wi, x, in_chunk0, rows, out_hw, kw, M, gflop, chosen):
    kw = dict(kw)
    kw.pop("variant", None)
    variants = (6, 5, 4, 2) if wi.Npad % 128 == 0 else (2, 7)      # 7 on 64 columns = the 256 x 64 tap-reuse ping-pong
    res = []
    for v in variants:
        for S in (1, 2, 4, 8, 16):
            kt = max(len(t) for _, _, t in wi.phases) * wi.KC
            if S > 1 and kt // S < 4:
                continue
            ws = torch.empty(len(wi.phases) * S * M * wi.Npad, device=DEV) if S > 1 else None
            try:
                launch = ig.make_launch(wi, x, in_chunk0, rows, out_hw, splitk=S, ws=ws, variant=v, **kw)
                ms = timed(launch)
            except RuntimeError as exc:
                print(json.dumps(dict(launch=label, variant=v, splitk=S, error=str(exc)[:100])), flush=True)
                continue
            res.append((ms, v, S))
            print(json.dumps(dict(launch=label, variant=v, splitk=S, ms=round(ms, 4), frac=round(gflop / ms / CEIL, 3),
                                  chosen=(v, S) == chosen)), flush=True)
    best = min(res)
    cur = [r for r in res if (r[1], r[2]) == chosen]
    print(json.dumps(dict(launch=label, best=dict(ms=round(best[0], 4), variant=best[1], splitk=best[2]),
                          engine_choice=dict(variant=chosen[0], splitk=chosen[1], ms=round(cur[0][0], 4) if cur else None))), flush=True)


def main():
    which = set(sys.argv[1:]) or {"window", "s2bwd", "small"}
    import bench
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1000.0, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    B, H, W, P_ = bench.B_PER_GPU, bench.H, bench.W, bench.PATCH
    step = PatchAttackStep(net, args, B, H, W, device=DEV, shared_patch=True, patch_hw=(P_, P_))
    tgt, ref, origins = bench.synthetic_batch(B, 1000, DEV)
    with torch.no_grad():
        target = -torch.cat([net(tgt[i:i + 1], ref[i:i + 1]) for i in range(B)])
    g = torch.Generator().manual_seed(7)
    patch0 = torch.rand(1, 3, P_, P_, generator=g).to(DEV)
    mask_p = bench.circle_mask(P_).expand(1, 3, P_, P_).contiguous().to(DEV)
    step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
    step.run(2)
    eng = step.eng
    if "window" in which:
        P = eng._wprefix
        wh, ww = P["hw"]
        B2 = 2 * B
        h2, w2, h4, w4, h8, w8 = wh // 2, ww // 2, wh // 4, ww // 4, wh // 8, ww // 8
        gz_c1 = ig.Planes(B2, h2, w2, 2, DEV)
        bias = lambda n: eng._conv(n).bias.detach().float().contiguous()
        plans = [("window conv2 fwd", "conv2", P["c1"], (h4, w4), (h4, w4), dict(out_planes=P["c2"], bias=bias("conv2"))),
                 ("window conv3 fwd", "conv3", P["c2"], (h8, w8), (h8, w8), dict(out_planes=P["c3"], bias=bias("conv3"))),
                 ("window conv3 bwd", "conv3_bwd", P["gz_c3"], (h8, w8), (h4, w4), dict(add=P["G_gw2"], mask=P["c2"], out_planes=P["gz_c2"])),
                 ("window conv2 bwd", "conv2_bwd", P["gz_c2"], (h4, w4), (h2, w2), dict(out_planes=gz_c1, mask=P["c1"])),
                 ("window conv1 bwd", "conv1_bwd", gz_c1, (h2 + 3, w2 + 2), (h2 + 3, w2 + 2), dict(out_f32=P["G_p"]))]
        for label, key, x, rows, out_hw, kw in plans:
            wi = P[key + "_wi"]
            d = P[key].desc
            M = B2 * rows[0] * rows[1]
            sweep(label, wi, x, 0, rows, out_hw, kw, M, wi.flops(M) / 1e9, (d.variant if d.variant else 2, d.splitk))
    for name, kind, tag, launch, gflop in eng.launch_table():
        s2 = kind == "bwd" and name in ("conv4", "conv5", "conv6")
        small = name in ("conv5_1", "conv6", "conv6_1", "deconv5", "conv5") and tag == "full"
        n64 = "n64" in which and name in ("deconv2", "conv_redir") and kind == "fwd"
        if not (("s2bwd" in which and s2) or ("small" in which and small and not s2) or n64) or tag in ("prefix", "window"):
            continue
        if not hasattr(eng, "replan"):
            break
        wi, x, in_chunk0, rows, out_hw, kw = eng.replan(kind, name, tag)
        d = launch.desc
        sweep(f"{name} {kind} ({tag})", wi, x, in_chunk0, rows, out_hw, kw, d.B * d.Hr * d.Wr, gflop, (d.variant if d.variant else 2, d.splitk))


if __name__ == "__main__":
    main()

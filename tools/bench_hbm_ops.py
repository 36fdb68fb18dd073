"""Per-kernel roofline figures of the HBM / L2-bound operators `north_star` names, at the BASELINE configs' sizes
(VERDICT r1 item 6): Resample2d and ChannelNorm (FlowNet2 @448x1024, config C5), RAFT's cost-volume lookup, alt_cuda_corr,
the SepConvGRU gate kernels and convex upsampling (RAFT @384x1280, config C3).  One JSON line per kernel: average duration
(HIP events on the launch stream), algorithmic bytes, GB/s, fraction of the 8 TB/s HBM peak."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from understanding_flow_robustness_amd import _lib as L

DEV = "cuda:0"
PEAK = 8000.0


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


def emit(kernel, config, ms, nbytes, note=""):
    gbs = nbytes / ms / 1e6
    print(json.dumps(dict(kernel=kernel, config=config, ms=round(ms, 4), algorithmic_bytes=int(nbytes), gbs=round(gbs, 1),
                          frac_of_hbm_peak=round(gbs / PEAK, 4), bound="hbm", note=note)), flush=True)


def main():
    only_resample = "--resample-only" in sys.argv
    lib = L.lib()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    st = L.stream
    # ---- FlowNet2 @448x1024 (C5: one pair per GPU, and 8 pairs)
    for B in (1, 8):
        H, W = 448, 1024
        img = torch.rand(B, 3, H, W, generator=g).to(DEV)
        out, gout = torch.empty_like(img), rnd(B, 3, H, W)
        cfg = f"FlowNet2 448x1024, {B} pair(s)"
        # two flow fields: 'smooth' = what a flow network emits (a 1/16-resolution field of +-16 px, bilinearly upsampled, plus
        # 0.25 px of noise); 'rough' = independent 4 px noise per pixel (worst case for locality)
        coarse = 16 * torch.randn(B, 2, H // 16, W // 16, generator=g)
        smooth = torch.nn.functional.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=False) + 0.25 * torch.randn(B, 2, H, W, generator=g)
        for kind, flow in (("smooth", smooth.to(DEV)), ("rough", (4 * torch.randn(B, 2, H, W, generator=g)).to(DEV))):
            gimg, gflow = torch.empty_like(img), torch.empty_like(flow)
            ms = timed(lambda: L.check(lib.ufr_resample2d_forward(L.ptr(img), L.ptr(flow), L.ptr(out), B, 3, H, W, H, W, 1, 1, st())))
            emit(f"resample2d_fwd ({kind} flow)", cfg, ms, (img.numel() + flow.numel() + out.numel()) * 4)
            ms = timed(lambda: L.check(lib.ufr_resample2d_backward(L.ptr(img), L.ptr(flow), L.ptr(gout), L.ptr(gimg), L.ptr(gflow), B, 3, H, W,
                                                                   H, W, 1, 1, st())))
            emit(f"resample2d_bwd (image + flow gradients, {kind} flow)", cfg, ms,
                 (img.numel() + flow.numel() + gout.numel() + gimg.numel() + gflow.numel()) * 4)
        for C in (3, 2):
            x = rnd(B, C, H, W)
            nrm, gn, gx = torch.empty(B, 1, H, W, device=DEV), rnd(B, 1, H, W), torch.empty_like(x)
            ms = timed(lambda: L.check(lib.ufr_channelnorm_forward(L.ptr(x), L.ptr(nrm), B, C, H, W, 2, st())))
            emit(f"channelnorm_fwd C={C}", cfg, ms, (x.numel() + nrm.numel()) * 4)
            ms = timed(lambda: L.check(lib.ufr_channelnorm_backward(L.ptr(x), L.ptr(nrm), L.ptr(gn), L.ptr(gx), B, C, H, W, 2, st())))
            emit(f"channelnorm_bwd C={C}", cfg, ms, (2 * x.numel() + 2 * nrm.numel()) * 4)
    if only_resample:
        return
    # ---- RAFT @384x1280 (C3): 48x160 cells, 4 pyramid levels, radius 4
    from understanding_flow_robustness_amd import alt_cuda_corr
    from understanding_flow_robustness_amd.flownets import raft as R
    from understanding_flow_robustness_amd.flownets.raft_corr import CorrBlock
    B, H, W, C, r = 1, 48, 160, 256, 4
    cfg = "RAFT 384x1280, 1 pair"
    f1, f2 = rnd(B, C, H, W), rnd(B, C, H, W)
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    coords = (torch.cat([xs, ys], 1) + 3.0 * torch.randn(B, 2, H, W, generator=g)).to(DEV)
    blk = CorrBlock(f1, f2, num_levels=4, radius=r)
    c_req = coords.clone().requires_grad_(False)
    ms = timed(lambda: blk(c_req))
    out_bytes = B * 324 * H * W * 4
    gathered = B * H * W * 4 * (2 * r + 2) ** 2 * 4                  # (2r+2)^2 volume cells per pixel and level
    emit("lookup_fwd (4 levels, one call)", cfg, ms, out_bytes + gathered, "bytes = 324-channel output + the (2r+2)^2 cells read per level")
    vols = [v.clone().requires_grad_(True) for v in blk.get_corr_pyramid()]
    from understanding_flow_robustness_amd.flownets.raft_corr import corr_lookup
    y = corr_lookup(vols, coords, r)
    gy = torch.randn_like(y)
    ms = timed(lambda: torch.autograd.grad(y, vols, gy, retain_graph=True))
    emit("lookup_bwd (4 levels, one call)", cfg, ms, out_bytes + 2 * gathered, "read-modify-write of the touched volume cells")
    f1n, c5 = f1.permute(0, 2, 3, 1).contiguous(), coords.permute(0, 2, 3, 1).reshape(B, 1, H, W, 2).contiguous()
    for lvl in range(4):
        f2n = rnd(B, H >> lvl, W >> lvl, C)
        cl = (c5 / 2 ** lvl).contiguous()
        (o,) = alt_cuda_corr.forward(f1n, f2n, cl, r)
        go = torch.randn_like(o)
        win_bytes = B * H * W * (2 * r + 2) ** 2 * C * 4             # every pixel reads its (2r+2)^2 x C window of fmap2 (L2-served)
        ms = timed(lambda: alt_cuda_corr.forward(f1n, f2n, cl, r))
        emit(f"altcorr_fwd level {lvl}", cfg, ms, (f1n.numel() + f2n.numel() + o.numel()) * 4,
             f"HBM-algorithmic bytes; the per-pixel windows re-read {win_bytes / 1e6:.0f} MB through L2")
        ms = timed(lambda: alt_cuda_corr.backward(f1n, f2n, cl, go, r))
        emit(f"altcorr_bwd level {lvl} (both adjoints)", cfg, ms, (2 * f1n.numel() + 2 * f2n.numel() + o.numel()) * 4, "")
    hcn = 128
    zr, h = rnd(B, 2 * hcn, H, W), rnd(B, hcn, H, W)
    out = R._GruGates.apply(zr.requires_grad_(True), h.requires_grad_(True))
    ms = timed(lambda: R._GruGates.apply(zr, h))
    emit("gru_gates_fwd", cfg, ms, (zr.numel() + h.numel() + 2 * h.numel()) * 4)
    gz, grh = torch.randn_like(out[0]), torch.randn_like(out[1])
    ms = timed(lambda: torch.autograd.grad(out, (zr, h), (gz, grh), retain_graph=True))
    emit("gru_gates_bwd", cfg, ms, (zr.numel() + h.numel() + 2 * h.numel() + zr.numel() + h.numel()) * 4)
    q, z = rnd(B, hcn, H, W).requires_grad_(True), torch.rand(B, hcn, H, W, generator=g).to(DEV).requires_grad_(True)
    ob = R._GruBlend.apply(q, z, h)
    ms = timed(lambda: R._GruBlend.apply(q, z, h))
    emit("gru_blend_fwd", cfg, ms, 4 * h.numel() * 4)
    gb = torch.randn_like(ob)
    ms = timed(lambda: torch.autograd.grad(ob, (q, z, h), gb, retain_graph=True))
    emit("gru_blend_bwd", cfg, ms, 7 * h.numel() * 4)
    fl, mk = rnd(B, 2, H, W).requires_grad_(True), rnd(B, 576, H, W).requires_grad_(True)
    up = R._ConvexUpsample.apply(fl, mk)
    ms = timed(lambda: R._ConvexUpsample.apply(fl, mk))
    emit("convex_up_fwd", cfg, ms, (mk.numel() + fl.numel() + up.numel()) * 4)
    gu = torch.randn_like(up)
    ms = timed(lambda: torch.autograd.grad(up, (fl, mk), gu, retain_graph=True))
    emit("convex_up_bwd", cfg, ms, (2 * mk.numel() + 2 * fl.numel() + up.numel()) * 4)


if __name__ == "__main__":
    main()

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
    """Average duration (ms) of `fn`: `iters` calls captured in ONE HIP graph and replayed between two events, so that the
    figure is kernel time (round 2 timed an eager loop: for 7 us kernels behind an autograd Function that measured the host,
    ~77 us per call, not the kernels)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(iters):
                fn()
    torch.cuda.current_stream().wait_stream(side)
    graph.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    graph.replay()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def emit(kernel, config, ms, nbytes, note=""):
    gbs = nbytes / ms / 1e6
    print(json.dumps(dict(kernel=kernel, config=config, ms=round(ms, 4), algorithmic_bytes=int(nbytes), gbs=round(gbs, 1),
                          frac_of_hbm_peak=round(gbs / PEAK, 4), bound="hbm", note=note)), flush=True)


def pwc_warp_rows(lib, g, st):
    """PWC-Net's warp and its adjoint at config C4's four warped levels (384x1280, 8 pairs): the scatter with float atomics (round 1)
    against the owner-computes form (round 5).  Flow = what the level's decoder sees: a smooth field of a few cells plus 0.25 of noise."""
    B = 8
    for k, C in ((5, 128), (4, 96), (3, 64), (2, 32)):
        H, W = 384 >> k, 1280 >> k
        cfg = f"PWC-Net 384x1280, {B} pairs, level {k} ({C} x {H} x {W})"
        x, go = torch.randn(B, C, H, W, generator=g).to(DEV), torch.randn(B, C, H, W, generator=g).to(DEV)
        up = lambda t: torch.nn.functional.interpolate(t, size=(H, W), mode="bilinear", align_corners=False)
        flows = {"smooth": up(3 * torch.randn(B, 2, max(H // 8, 1), max(W // 8, 1), generator=g)) + 0.25 * torch.randn(B, 2, H, W, generator=g),
                 "gentle": up(2 * torch.randn(B, 2, max(H // 24, 1), max(W // 24, 1), generator=g)),
                 "zero": torch.zeros(B, 2, H, W)}
        out, gx, gf = torch.empty_like(x), torch.empty_like(x), torch.empty_like(flow := flows["smooth"].to(DEV))
        for kind in ("gentle", "zero"):
            fl = flows[kind].to(DEV)
            nb = int(lib.ufr_pwc_warp_backward_workspace_bytes(B, H, W))
            ws = torch.empty((nb + 15) // 16 * 4, dtype=torch.int32, device=DEV)
            nbytes = (3 * x.numel() + 2 * fl.numel()) * 4
            ms = timed(lambda: L.check(lib.ufr_pwc_warp_backward(L.ptr(x), L.ptr(fl), L.ptr(go), L.ptr(gx), L.ptr(gf), B, C, H, W, st())))
            emit(f"pwc_warp_bwd, scatter ({kind} flow)", cfg, ms, nbytes)
            ms = timed(lambda: L.check(lib.ufr_pwc_warp_backward_owner(L.ptr(x), L.ptr(fl), L.ptr(go), L.ptr(gx), L.ptr(gf), L.ptr(ws), nb, B, C, H, W,
                                                                       st())))
            emit(f"pwc_warp_bwd, owner-computes ({kind} flow)", cfg, ms, nbytes)
        nbytes = (3 * x.numel() + 2 * flow.numel()) * 4                 # x, grad_out in; grad_x out; flow in, grad_flow out
        ms = timed(lambda: L.check(lib.ufr_pwc_warp_forward(L.ptr(x), L.ptr(flow), L.ptr(out), B, C, H, W, st())))
        emit("pwc_warp_fwd", cfg, ms, (2 * x.numel() + flow.numel()) * 4)
        ms = timed(lambda: L.check(lib.ufr_pwc_warp_backward(L.ptr(x), L.ptr(flow), L.ptr(go), L.ptr(gx), L.ptr(gf), B, C, H, W, st())))
        emit("pwc_warp_bwd, scatter with float atomics (+ the zero fill)", cfg, ms, nbytes)
        nb = int(lib.ufr_pwc_warp_backward_workspace_bytes(B, H, W))
        ws = torch.empty((nb + 15) // 16 * 4, dtype=torch.int32, device=DEV)
        ms = timed(lambda: L.check(lib.ufr_pwc_warp_backward_owner(L.ptr(x), L.ptr(flow), L.ptr(go), L.ptr(gx), L.ptr(gf), L.ptr(ws), nb, B, C, H, W,
                                                                   st())))
        emit("pwc_warp_bwd, owner-computes (the default)", cfg, ms, nbytes)


def main():
    only_resample = "--resample-only" in sys.argv
    lib = L.lib()
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(DEV)
    st = L.stream
    if "--warp-only" in sys.argv:
        return pwc_warp_rows(lib, g, st)
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
        # 'gentle' = a 1/64-resolution field of +-8 px, upsampled: gradients of ~0.2 px per px, what most of a real frame looks like
        gentle = torch.nn.functional.interpolate(8 * torch.randn(B, 2, H // 64, W // 64, generator=g), size=(H, W), mode="bilinear", align_corners=False)
        for kind, flow in (("gentle", gentle.to(DEV)), ("smooth", smooth.to(DEV)), ("rough", (4 * torch.randn(B, 2, H, W, generator=g)).to(DEV))):
            gimg, gflow = torch.empty_like(img), torch.empty_like(flow)
            ms = timed(lambda: L.check(lib.ufr_resample2d_forward(L.ptr(img), L.ptr(flow), L.ptr(out), B, 3, H, W, H, W, 1, 1, st())))
            emit(f"resample2d_fwd ({kind} flow)", cfg, ms, (img.numel() + flow.numel() + out.numel()) * 4)
            ms = timed(lambda: L.check(lib.ufr_resample2d_backward(L.ptr(img), L.ptr(flow), L.ptr(gout), L.ptr(gimg), L.ptr(gflow), B, 3, H, W,
                                                                   H, W, 1, 1, st())))
            emit(f"resample2d_bwd, LDS-privatised scatter (image + flow gradients, {kind} flow)", cfg, ms,
                 (img.numel() + flow.numel() + gout.numel() + gimg.numel() + gflow.numel()) * 4)
            nb = int(lib.ufr_resample2d_backward_workspace_bytes(B, H, W))
            ws = torch.empty((nb + 15) // 16 * 4, dtype=torch.int32, device=DEV)
            ms = timed(lambda: L.check(lib.ufr_resample2d_backward_owner(L.ptr(img), L.ptr(flow), L.ptr(gout), L.ptr(gimg), L.ptr(gflow), L.ptr(ws),
                                                                         nb, B, 3, H, W, st())))
            emit(f"resample2d_bwd, owner-computes (the default; image + flow gradients, {kind} flow)", cfg, ms,
                 (img.numel() + flow.numel() + gout.numel() + gimg.numel() + gflow.numel()) * 4)
            if kind == "gentle":
                # the FLOOR of that adjoint on this box: the same bytes -- image, flow and output gradient in, image and
                # flow gradients out -- moved by two streaming kernels that compute nothing (gimg = img + gout; gflow = flow)
                def floor():
                    torch.add(img, gout, out=gimg)
                    gflow.copy_(flow)
                ms = timed(floor)
                emit("resample2d_bwd FLOOR: the adjoint's bytes moved by two streaming kernels (gimg = img + gout; gflow = flow)", cfg, ms,
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
    B, H, W, C_, r = 1, 48, 160, 256, 4
    cfg = "RAFT 384x1280, 1 pair"
    f1, f2 = rnd(B, C_, H, W), rnd(B, C_, H, W)
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    coords = (torch.cat([xs, ys], 1) + 3.0 * torch.randn(B, 2, H, W, generator=g)).to(DEV)
    blk = CorrBlock(f1, f2, num_levels=4, radius=r)
    c_req = coords.clone().requires_grad_(False)
    ms = timed(lambda: blk(c_req))
    out_bytes = B * 324 * H * W * 4
    gathered = B * H * W * 4 * (2 * r + 2) ** 2 * 4                  # (2r+2)^2 volume cells per pixel and level
    emit("lookup_fwd (4 levels, one call)", cfg, ms, out_bytes + gathered, "bytes = 324-channel output + the (2r+2)^2 cells read per level")
    import ctypes as C
    from understanding_flow_robustness_amd.flownets.raft_corr import _levels_struct, _pyramid_struct
    vols = [v.contiguous() for v in blk.get_corr_pyramid()]
    gvols = [torch.zeros_like(v) for v in vols]
    gy = torch.randn(B, 324, H, W, generator=g).to(DEV)
    pyr = _pyramid_struct(vols, gvols)
    ms = timed(lambda: L.check(lib.ufr_corr_lookup_backward(C.byref(pyr), L.ptr(coords), L.ptr(gy), B, H, W, r, st())))
    emit("lookup_bwd (4 levels, one call)", cfg, ms, out_bytes + 2 * gathered, "read-modify-write of the touched volume cells")
    f1n, c5 = f1.permute(0, 2, 3, 1).contiguous(), coords.permute(0, 2, 3, 1).reshape(B, 1, H, W, 2).contiguous()
    for lvl in range(4):
        f2n = rnd(B, H >> lvl, W >> lvl, C_)
        cl = (c5 / 2 ** lvl).contiguous()
        (o,) = alt_cuda_corr.forward(f1n, f2n, cl, r)
        go = torch.randn_like(o)
        win_bytes = B * H * W * (2 * r + 2) ** 2 * C_ * 4             # every pixel reads its (2r+2)^2 x C window of fmap2 (L2-served)
        ms = timed(lambda: alt_cuda_corr.forward(f1n, f2n, cl, r))
        emit(f"altcorr_fwd level {lvl}", cfg, ms, (f1n.numel() + f2n.numel() + o.numel()) * 4,
             f"HBM-algorithmic bytes; the per-pixel windows re-read {win_bytes / 1e6:.0f} MB through L2")
        ms = timed(lambda: alt_cuda_corr.backward(f1n, f2n, cl, go, r))
        emit(f"altcorr_bwd level {lvl} (both adjoints)", cfg, ms, (2 * f1n.numel() + 2 * f2n.numel() + o.numel()) * 4, "")
    # the model's form: all four levels of a lookup in one launch per direction on the fp32 matrix cores (raft_altcorr_mfma.hip)
    from understanding_flow_robustness_amd.flownets.raft_corr import AltCorrPyramidFunction
    f2s = [rnd(B, H >> lvl, W >> lvl, C_) for lvl in range(4)]
    scale = 1.0 / 16.0
    o4 = AltCorrPyramidFunction.apply(f1n, coords, r, scale, None, *f2s)
    nb_in = (f1n.numel() + sum(f.numel() for f in f2s)) * 4
    ms = timed(lambda: AltCorrPyramidFunction.apply(f1n, coords, r, scale, None, *f2s))
    emit("altcorr_mfma_fwd (4 levels, one launch)", cfg, ms, nb_in + o4.numel() * 4, "fmap1 + the four fmap2 levels in, [B,324,H,W] out")
    g1, g2s, go4 = torch.empty_like(f1n), [torch.empty_like(f) for f in f2s], torch.randn_like(o4)
    wsb = torch.empty(lib.ufr_altcorr_pyramid_workspace_bytes(B, H, W, C_, r, 4), dtype=torch.uint8, device=DEV)
    lv = _levels_struct(f2s, g2s)
    ms = timed(lambda: L.check(lib.ufr_altcorr_pyramid_backward(L.ptr(f1n), C.byref(lv), L.ptr(coords), L.ptr(go4), L.ptr(g1), L.ptr(wsb), B, H, W,
                                                                C_, r, scale, 0, st())))
    emit("altcorr_mfma backward (prepass + d/d fmap1 + level sum + d/d fmap2, 4 levels)", cfg, ms, 2 * nb_in + o4.numel() * 4, "")
    hcn = 128
    zr, h = rnd(B, 2 * hcn, H, W), rnd(B, hcn, H, W)
    z, rh = torch.empty_like(h), torch.empty_like(h)
    ms = timed(lambda: L.check(lib.ufr_gru_gates_forward(L.ptr(zr), L.ptr(h), L.ptr(z), L.ptr(rh), B, hcn, H * W, hcn * H * W, st())))
    emit("gru_gates_fwd", cfg, ms, (zr.numel() + h.numel() + 2 * h.numel()) * 4)
    gz, grh, gzr, gh = rnd(B, hcn, H, W), rnd(B, hcn, H, W), torch.empty_like(zr), torch.empty_like(h)
    ms = timed(lambda: L.check(lib.ufr_gru_gates_backward(L.ptr(zr), L.ptr(h), L.ptr(gz), L.ptr(grh), L.ptr(gzr), L.ptr(gh), B, hcn, H * W,
                                                          hcn * H * W, st())))
    emit("gru_gates_bwd", cfg, ms, (zr.numel() + h.numel() + 2 * h.numel() + zr.numel() + h.numel()) * 4)
    q, zz, ho = rnd(B, hcn, H, W), torch.rand(B, hcn, H, W, generator=g).to(DEV), torch.empty_like(h)
    ms = timed(lambda: L.check(lib.ufr_gru_blend_forward(L.ptr(q), L.ptr(zz), L.ptr(h), L.ptr(ho), h.numel(), st())))
    emit("gru_blend_fwd", cfg, ms, 4 * h.numel() * 4)
    gb, gq, gz2, gh2 = rnd(B, hcn, H, W), torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
    ms = timed(lambda: L.check(lib.ufr_gru_blend_backward(L.ptr(q), L.ptr(zz), L.ptr(h), L.ptr(gb), L.ptr(gq), L.ptr(gz2), L.ptr(gh2), h.numel(), st())))
    emit("gru_blend_bwd", cfg, ms, 7 * h.numel() * 4)
    # the engine's chunk-major forms (csrc/raft_update.hip): values in place, r*h / h' as planes
    from understanding_flow_robustness_amd import igemm as ig
    M = B * H * W
    ZR, Q, Pb, Gz, Gh, Grh = ig.GradSum(B, H, W, 8, DEV), ig.GradSum(B, H, W, 4, DEV), ig.Planes(B, H, W, 16, DEV), ig.GradSum(B, H, W, 4, DEV), \
        ig.GradSum(B, H, W, 4, DEV), ig.GradSum(B, H, W, 4, DEV)
    gzq, gzrp = ig.Planes(B, H, W, 4, DEV), ig.Planes(B, H, W, 8, DEV)
    ZR.t.normal_(); Q.t.normal_(); Pb.t.normal_(); Gh.t.normal_(); Grh.t.normal_()
    ms = timed(lambda: L.check(lib.ufr_gru_gates_cm_forward(L.ptr(ZR.t), L.ptr(Pb.t), Pb.plane_stride, 0, L.ptr(Pb.t), Pb.plane_stride, 12, M, 4, st())))
    emit("gates_fwd_kernel (chunk-major: zr in place, r*h planes)", cfg, ms, M * 128 * (8 + 8 + 6 + 6))
    ms = timed(lambda: L.check(lib.ufr_gru_blend_cm_forward(L.ptr(Q.t), L.ptr(ZR.t), L.ptr(Pb.t), Pb.plane_stride, 0, L.ptr(Pb.t), Pb.plane_stride, 4, M,
                                                            4, st())))
    emit("blend_fwd_kernel (chunk-major)", cfg, ms, M * 128 * (4 + 4 + 4 + 6 + 6))
    ms = timed(lambda: L.check(lib.ufr_gru_blend_cm_backward(L.ptr(Q.t), L.ptr(ZR.t), L.ptr(Pb.t), Pb.plane_stride, 0, L.ptr(Gh.t), L.ptr(gzq.t),
                                                             gzq.plane_stride, 0, L.ptr(Gz.t), L.ptr(Grh.t), M, 4, None, st())))
    emit("blend_bwd_kernel (chunk-major)", cfg, ms, M * 128 * (4 + 4 + 6 + 4 + 6 + 4 + 4))
    ms = timed(lambda: L.check(lib.ufr_gru_gates_cm_backward(L.ptr(ZR.t), L.ptr(Pb.t), Pb.plane_stride, 0, L.ptr(Gz.t), L.ptr(Grh.t), L.ptr(gzrp.t),
                                                             gzrp.plane_stride, 0, L.ptr(Gh.t), M, 4, 0, None, st())))
    emit("gates_bwd_kernel (chunk-major)", cfg, ms, M * 128 * (8 + 6 + 4 + 4 + 12 + 8))
    fl, mk = rnd(B, 2, H, W), rnd(B, 576, H, W)
    up = torch.empty(B, 2, 8 * H, 8 * W, device=DEV)
    ms = timed(lambda: L.check(lib.ufr_convex_upsample_forward(L.ptr(fl), L.ptr(mk), L.ptr(up), B, H, W, st())))
    emit("convex_up_fwd", cfg, ms, (mk.numel() + fl.numel() + up.numel()) * 4)
    gu, gfl, gmk, wsu = torch.randn_like(up), torch.empty_like(fl), torch.empty_like(mk), torch.empty(B, 2, 9, H, W, device=DEV)
    ms = timed(lambda: L.check(lib.ufr_convex_upsample_backward(L.ptr(fl), L.ptr(mk), L.ptr(gu), L.ptr(gfl), L.ptr(gmk), L.ptr(wsu), B, H, W, st())))
    emit("convex_up_bwd", cfg, ms, (2 * mk.numel() + 2 * fl.numel() + up.numel()) * 4)


if __name__ == "__main__":
    main()

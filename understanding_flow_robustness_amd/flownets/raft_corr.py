"""RAFT cost-volume blocks on the gfx950 kernels (csrc/raft_corr.hip).

Mirrors models/raft/corr.py: `CorrBlock` (all-pairs volume + 4-level pyramid + windowed bilinear
lookup, :26-106) and `AlternateCorrBlock` (on-the-fly correlation through alt_cuda_corr, :109-137).
The all-pairs product is a plain library GEMM (torch.matmul -> hipBLASLt); the lookup -- 48
`grid_sample` calls per RAFT forward in the reference -- is one fused kernel per call here.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
import torch.nn.functional as F

from .. import _lib as L
from ..alt_cuda_corr import AltCorrFunction


def _pyramid_struct(vols, grads=None) -> L.Pyramid:
    pyr = L.Pyramid()
    pyr.num_levels = len(vols)
    for i, v in enumerate(vols):
        pyr.vol[i] = v.data_ptr()
        pyr.grad_vol[i] = grads[i].data_ptr() if grads is not None else None
        pyr.Hl[i], pyr.Wl[i] = int(v.shape[-2]), int(v.shape[-1])
    return pyr


class CorrLookupFunction(torch.autograd.Function):
    """out[B, L*(2r+1)^2, H, W] = windowed bilinear samples of every pyramid level (corr.py:72-96)."""

    @staticmethod
    def forward(ctx, coords, radius, shared, *vols):
        """`shared`: None, or a `_SharedGrad` through which all lookups of one pyramid (RAFT's 12
        iterations) accumulate their adjoints into ONE set of buffers: the kernel's `+=` replaces
        12 zero-fills and 11 full-volume additions of autograd's gradient accumulation."""
        L.require_hip(coords, "coords")
        coords = coords.contiguous()
        vols = tuple(v.contiguous() for v in vols)
        for v in vols:
            L.require_hip(v, "corr pyramid level")
            if v.dtype != torch.float32:
                raise RuntimeError("corr pyramid must be float32")
        B, two, H1, W1 = coords.shape
        if two != 2 or any(v.shape[0] != B * H1 * W1 for v in vols):
            raise RuntimeError("coords must be [B,2,H,W] and every level [B*H*W,1,Hl,Wl]")
        rd = 2 * int(radius) + 1
        with torch.cuda.device(coords.device):
            out = torch.empty((B, len(vols) * rd * rd, H1, W1), dtype=torch.float32, device=coords.device)
            pyr = _pyramid_struct(vols)
            L.check(L.lib().ufr_corr_lookup_forward(C.byref(pyr), L.ptr(coords), L.ptr(out), B, H1, W1,
                                                    int(radius), L.stream()), "corr lookup forward")
        ctx.save_for_backward(coords, *vols)
        ctx.radius = int(radius)
        ctx.shared = shared
        if shared is not None:
            shared.pending += 1
        return out

    @staticmethod
    def backward(ctx, grad_out):
        coords, *vols = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        B, _, H1, W1 = coords.shape
        shared = ctx.shared
        with torch.cuda.device(coords.device):
            if shared is None:
                grads = [torch.zeros_like(v) for v in vols]
            else:
                if shared.acc is None:
                    shared.acc = [torch.zeros_like(v) for v in vols]
                grads = shared.acc
            pyr = _pyramid_struct(vols, grads)
            L.check(L.lib().ufr_corr_lookup_backward(C.byref(pyr), L.ptr(coords), L.ptr(grad_out), B, H1,
                                                     W1, ctx.radius, L.stream()), "corr lookup backward")
        if shared is not None:
            shared.pending -= 1
            if shared.pending > 0:          # more adjoints to come: hand autograd nothing yet
                return (None, None, None, *([None] * len(vols)))
            shared.acc = None               # the last adjoint delivers the accumulated volumes
        return (None, None, None, *grads)   # coords are detached every RAFT iteration (raft.py:190)


class _SharedGrad:
    """Accumulation state shared by every lookup of one CorrBlock (see CorrLookupFunction.forward).
    Valid when every lookup's output reaches the loss (true for RAFT: the hidden state chains the
    iterations), because the last adjoint to run is the one that returns the sum."""

    def __init__(self):
        self.acc, self.pending = None, 0


def corr_lookup(pyramid, coords, radius, shared=None):
    return CorrLookupFunction.apply(coords, radius, shared, *pyramid)


_ALL_PAIRS_PLANES = L.LruDict(8)


def _all_pairs_planes(device, H, W, chunks):
    """The two operand buffers of the all-pairs launch, one pair per shape (stream-ordered reuse: a launch reads them before the
    next load overwrites them)."""
    from .. import igemm as ig
    key = (device, H, W, chunks)
    if key not in _ALL_PAIRS_PLANES:
        _ALL_PAIRS_PLANES[key] = (ig.Planes(1, H, W, chunks, device), ig.Planes(1, H, W, chunks, device))
    return _ALL_PAIRS_PLANES.get(key)


_ALL_PAIRS_ADJOINT = L.LruDict(4)


def _all_pairs_adjoint(fmap1, fmap2, g, scale, needs):
    """Both adjoint products of the all-pairs correlation on the igemm (see AllPairsCorrFunction); g [B, HW, HW] contiguous."""
    from .. import igemm as ig
    B, C_, H, W = fmap1.shape
    HW, dev = H * W, fmap1.device
    key = (dev, H, W, C_)
    st = _ALL_PAIRS_ADJOINT.get(key)
    if st is None:
        st = _ALL_PAIRS_ADJOINT[key] = dict(A=ig.Planes(1, H, W, HW // 32, dev),            # the gradient as the activation
                                            Wf=ig.Planes(1, 1, C_, HW // 32, dev),           # a feature map [C, HW] as the weights
                                            out=ig.GradSum(1, H, W, C_ // 32, dev))
        wi = ig.planes_as_weights(st["Wf"])
        S = ig.splitk_for(HW, wi.Npad, wi.KC, 1, bm=256, target=256, min_ktiles=4)
        wi_variant, S = ig.tuned(wi, HW, dict(out_f32=st["out"]), 6, S)
        st["ws"] = torch.empty(max(1, S * HW * wi.Npad), dtype=torch.float32, device=dev) if S > 1 else None
        st["launch"] = ig.make_launch(wi, st["A"], 0, (H, W), (H, W), out_f32=st["out"], splitk=S, ws=st["ws"], variant=wi_variant)
    A, Wf, out, launch = st["A"], st["Wf"], st["out"], st["launch"]
    f1, f2 = fmap1.detach().contiguous().view(B, C_, HW), fmap2.detach().contiguous().view(B, C_, HW)
    g1 = torch.empty_like(fmap1) if needs[0] else None
    g2 = torch.empty_like(fmap2) if needs[1] else None
    with torch.cuda.device(dev):
        for b in range(B):
            if needs[0]:
                A.load_rowmajor(g[b])                                  # A[p][q] = g[p, q]
                Wf.load_rowmajor(f2[b])                                # W[c][q] = fmap2[c, q]
                launch()
                out.to_nchw(C_, 0, scale=scale, slope=1.0, out=g1[b:b + 1])
            if needs[1]:
                A.load_nchw(g[b].view(1, HW, H, W))                    # A[q][p] = g[p, q]: channels = g's rows
                Wf.load_rowmajor(f1[b])                                # W[c][p] = fmap1[c, p]
                launch()
                out.to_nchw(C_, 0, scale=scale, slope=1.0, out=g2[b:b + 1])
    return g1, g2


_ALT_DENSE = L.LruDict(4)


def alt_dense_adjoint_served(f1, f2s, radius) -> bool:
    """The dense adjoint of AlternateCorrBlock's lookups (below) serves HIP float32 maps whose channel count the igemm takes as a
    weight image (a multiple of 64) and RAFT's radii; UFR_ALTCORR_DENSE_ADJOINT=0 keeps the on-the-fly adjoint kernels."""
    return (f1.is_cuda and f1.dtype == torch.float32 and f1.shape[3] % 64 == 0 and int(radius) in (3, 4) and 1 <= len(f2s) <= 4
            and os.environ.get("UFR_ALTCORR_DENSE_ADJOINT", "1") != "0")


def alt_dense_volumes(f1, f2s):
    """Zeroed gradient volumes of the lookups' windows, level l: [B * H1 * W1, 1, H_l, W_l] (the layout `ufr_corr_lookup_backward` adds into)."""
    B, H1, W1, _ = f1.shape
    return [torch.zeros(B * H1 * W1, 1, f.shape[1], f.shape[2], dtype=torch.float32, device=f1.device) for f in f2s]


def alt_dense_adjoint(f1, f2s, g_vols, scale):
    """d / d fmap1 and d / d fmap2_l of ALL lookups of one AlternateCorrBlock at once (round 6).  Every lookup's adjoint is a sparse
    matrix G_l[p, q] (pixel p, window point q of level l) times a feature map: d f1[p] = sum_l sum_q G_l[p, q] f2_l[q],
    d f2_l[q] = sum_p G_l[p, q] f1[p] (corr.py:121-137 is linear in both maps, coords are detached, raft.py:190).  RAFT's 12 lookups
    share the maps, so their G_l are ADDED first -- `lookup_bwd_tiled` scatters a lookup's window adjoints into dense volumes in 31 us
    (one writer per pixel slice, no atomics) -- and the products are taken ONCE, as 2 x levels launches of the hand-written igemm with the
    volume as the activation and a feature map's planes as the weight image (`_all_pairs_adjoint`'s form, per level and rectangular).
    That replaces 12 x (pre-pass + two gather-GEMM adjoints + two fixed-order sums) = 3.2 ms of kernel time per RAFT iteration by
    12 x 0.04 + ~0.8 ms, at the price of 313 MB of volume gradient per pair -- nothing on a 288 GB part.  f1 [B,H1,W1,C], f2s[l] [B,Hl,Wl,C]
    (NHWC float32), g_vols[l] [B*H1*W1, 1, Hl, Wl]; returns (g_f1, [g_f2_l]) in NHWC, multiplied by `scale`."""
    from .. import igemm as ig
    B, H1, W1, C_ = f1.shape
    HW1, dev = H1 * W1, f1.device
    shapes = tuple((int(f.shape[1]), int(f.shape[2])) for f in f2s)
    key = (dev, H1, W1, C_, shapes)
    st = _ALT_DENSE.get(key)
    if st is None:
        kc1 = (HW1 + 31) // 32
        st = dict(W1=ig.Planes(1, 1, C_, kc1, dev), out1=ig.GradSum(1, H1, W1, C_ // 32, dev), levels=[])
        big = None                                      # level 0's two activation buffers have one shape: one allocation
        for l, (Hl, Wl) in enumerate(shapes):
            kcl = (Hl * Wl + 31) // 32
            A1 = ig.Planes(1, H1, W1, kcl, dev)         # G_l as [pixels p][window points q]
            if (Hl, Wl) == (H1, W1):
                big = A2 = A1
            else:
                A2 = ig.Planes(1, Hl, Wl, kc1, dev)     # G_l^T as [points q][pixels p]
            Wf = ig.Planes(1, 1, C_, kcl, dev)          # fmap2_l as the weight image [C][q]
            out2 = ig.GradSum(1, Hl, Wl, C_ // 32, dev)
            lv = dict(A1=A1, A2=A2, Wf=Wf, out2=out2)
            for name, A, W_, out, rows, extra in (("p1", A1, Wf, st["out1"], (H1, W1), dict(add=st["out1"]) if l else {}),
                                                  ("p2", A2, st["W1"], out2, (Hl, Wl), {})):
                wi = ig.planes_as_weights(W_)
                M = rows[0] * rows[1]
                kw = dict(out_f32=out, **extra)
                S = ig.splitk_for(M, wi.Npad, wi.KC, 1, bm=256, target=256, min_ktiles=4)
                variant, S = ig.tuned(wi, M, kw, 6, S, rows=rows)
                ws = torch.empty(max(1, S * M * wi.Npad), dtype=torch.float32, device=dev) if S > 1 else None
                lv[name] = ig.make_launch(wi, A, 0, rows, rows, splitk=S, ws=ws, variant=variant, **kw)
            st["levels"].append(lv)
        _ALT_DENSE[key] = st
    g_f1 = torch.empty_like(f1)
    g_f2 = [torch.empty_like(f) for f in f2s]
    with torch.cuda.device(dev):
        for b in range(B):
            st["W1"].load_rowmajor(f1[b].reshape(HW1, C_).t().contiguous())                 # W[c][p] = fmap1[p, c]
            for l, ((Hl, Wl), lv) in enumerate(zip(shapes, st["levels"])):
                G = g_vols[l][b * HW1:(b + 1) * HW1]
                lv["Wf"].load_rowmajor(f2s[l][b].reshape(Hl * Wl, C_).t().contiguous())       # W[c][q] = fmap2_l[q, c]
                lv["A1"].load_rowmajor(G.view(HW1, Hl * Wl))                                  # A[p][q] = G_l[p, q]
                lv["p1"]()                                                                    # out1[p, c] (+)= sum_q G_l[p, q] fmap2_l[q, c]
                lv["A2"].load_nchw(G.view(1, HW1, Hl, Wl))                                    # A[q][p] = G_l[p, q]: channels = G's rows
                lv["p2"]()                                                                    # out2[q, c] = sum_p G_l[p, q] fmap1[p, c]
                torch.mul(lv["out2"].t.permute(1, 0, 2).reshape(Hl, Wl, C_), scale, out=g_f2[l][b])
            torch.mul(st["out1"].t.permute(1, 0, 2).reshape(H1, W1, C_), scale, out=g_f1[b])
    return g_f1, g_f2


class AllPairsCorrFunction(torch.autograd.Function):
    """corr[b, p, q] = <fmap1[b, :, p], fmap2[b, :, q]> / sqrt(C) (models/raft/corr.py:57-64) on the hand-written igemm
    (csrc/igemm.hip): per frame pair ONE 1x1 launch whose activation is fmap1's planes and whose "weight image" is fmap2's planes
    (the two layouts coincide, igemm.planes_as_weights), float32-accurate on the bf16 matrix cores, the row-major volume written
    by the epilogue.  The 1/sqrt(C) goes into fmap1's planes (exact for C = 256: a power of two).
    Backward (round 5) = the two products with the volume's gradient g[p, q] on the same kernel, per frame pair:
        d fmap1[c, p] = scale * sum_q g[p, q] fmap2[c, q]    rows p, reduction over g's COLUMNS: g through `Planes.load_rowmajor`,
        d fmap2[c, q] = scale * sum_p g[p, q] fmap1[c, p]    rows q, reduction over g's ROWS: g read as [channels p][pixels q],
    the feature maps [C, HW] as the weight image (row-major -> planes with C "pixels": the layouts coincide, planes_as_weights),
    float32 sums out, split-K with the fixed-order reduction.  UFR_ALLPAIRS_ADJOINT=0 keeps the library GEMMs (A/B only)."""

    @staticmethod
    def supported(fmap1, fmap2) -> bool:
        B, C_, H, W = fmap1.shape
        return (fmap1.is_cuda and fmap2.is_cuda and fmap1.device == fmap2.device and fmap1.dtype == torch.float32
                and fmap2.dtype == torch.float32 and fmap1.shape == fmap2.shape and C_ % 32 == 0 and (H * W) % 128 == 0)

    @staticmethod
    def forward(ctx, fmap1, fmap2):
        from .. import igemm as ig
        B, C_, H, W = fmap1.shape
        HW = H * W
        scale = 1.0 / math.sqrt(C_)
        # folded into fmap1 BEFORE the three-way bf16 split only where that is exact (a power of two: C = 64, 256, ...); for
        # raft-small's C = 128 the volume is scaled afterwards, like the reference's matmul(...) / sqrt(dim) (corr.py:63)
        fold = math.frexp(scale)[0] == 0.5
        out = torch.empty(B, HW, HW, dtype=torch.float32, device=fmap1.device)
        f1, f2 = fmap1.detach().contiguous(), fmap2.detach().contiguous()
        with torch.cuda.device(fmap1.device):
            p1, p2 = _all_pairs_planes(fmap1.device, H, W, C_ // 32)         # load_nchw writes every chunk: no zero fill per call
            for b in range(B):
                p1.load_nchw(f1[b:b + 1], 0, scale=scale if fold else 1.0)
                p2.load_nchw(f2[b:b + 1], 0)
                ig.make_launch(ig.planes_as_weights(p2), p1, 0, (H, W), (H, W), out_rowmajor=(out, b * HW * HW, HW), variant=6)()
            if not fold:
                out.mul_(scale)
        ctx.save_for_backward(fmap1, fmap2)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, g):
        fmap1, fmap2 = ctx.saved_tensors
        B, C_, H, W = fmap1.shape
        g = g.reshape(B, H * W, H * W)
        g1 = g2 = None
        if (AllPairsCorrFunction.supported(fmap1, fmap2) and g.is_cuda and g.dtype == torch.float32
                and os.environ.get("UFR_ALLPAIRS_ADJOINT", "1") != "0"):
            return _all_pairs_adjoint(fmap1, fmap2, g.contiguous(), ctx.scale, ctx.needs_input_grad)
        if ctx.needs_input_grad[0]:
            g1 = (torch.matmul(fmap2.reshape(B, C_, -1), g.transpose(1, 2)) * ctx.scale).view_as(fmap1)
        if ctx.needs_input_grad[1]:
            g2 = (torch.matmul(fmap1.reshape(B, C_, -1), g) * ctx.scale).view_as(fmap2)
        return g1, g2


class CorrBlock:
    """models/raft/corr.py:26-106 (all-pairs branch; `compute_spatial` is a visualisation aid)."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, share_grad=False):
        """`share_grad=True` is RAFT's own opt-in (flownets/raft.py): every lookup of its loop reaches the loss and coords are
        detached (raft.py:190), so the lookups' feature-map adjoints accumulate in one buffer and the last adjoint delivers the
        sum.  A public caller gets one independent autograd node per lookup (the default)."""
        self.num_levels, self.radius = num_levels, radius
        self._shared = _SharedGrad() if share_grad else None
        corr = CorrBlock.corr(fmap1, fmap2)
        batch, h1, w1, dim, h2, w2 = corr.shape
        corr = corr.reshape(batch * h1 * w1, dim, h2, w2)
        self.corr_pyramid = [corr]
        for _ in range(num_levels - 1):
            corr = F.avg_pool2d(corr, 2, stride=2)
            self.corr_pyramid.append(corr)

    def get_corr_pyramid(self):
        return self.corr_pyramid

    def __call__(self, coords):
        needs_grad = torch.is_grad_enabled() and any(v.requires_grad for v in self.corr_pyramid)
        return corr_lookup(self.corr_pyramid, coords, self.radius, self._shared if needs_grad else None)

    @staticmethod
    def corr(fmap1, fmap2):
        batch, dim, ht, wd = fmap1.shape
        if AllPairsCorrFunction.supported(fmap1, fmap2) and os.environ.get("UFR_ENGINE", "1") == "1":
            return AllPairsCorrFunction.apply(fmap1, fmap2).view(batch, ht, wd, 1, ht, wd)      # the hand-written igemm
        corr = torch.matmul(fmap1.view(batch, dim, ht * wd).transpose(1, 2), fmap2.view(batch, dim, ht * wd))
        return corr.view(batch, ht, wd, 1, ht, wd) / math.sqrt(dim)


def _levels_struct(f2s, grads=None) -> L.AltCorrLevels:
    lv = L.AltCorrLevels()
    lv.num_levels = len(f2s)
    for i, f in enumerate(f2s):
        lv.fmap2[i] = f.data_ptr()
        lv.fmap2_grad[i] = grads[i].data_ptr() if grads is not None else None
        lv.H2[i], lv.W2[i] = int(f.shape[1]), int(f.shape[2])
        lv.coord_scale[i] = 1.0 / 2 ** i                            # corr.py:126
    return lv


class AltCorrPlanes:
    """fmap1 and the fmap2 pyramid of one AlternateCorrBlock as bf16 split planes [3][C / 32][pixels][32] (csrc/igemm.hip's
    activation layout), made ONCE -- the maps are constant over RAFT's 12 lookups (models/raft/corr.py:111-119) -- for the lookup
    on the bf16 matrix cores with float32 accuracy (csrc/raft_altcorr_planes.hip, round 6).  `served()` says whether the kernel
    takes the shape (C 128 / 256, radius 3 / 4, at most four levels)."""

    @staticmethod
    def served(f1, f2s, radius) -> bool:
        return (f1.is_cuda and f1.dtype == torch.float32 and f1.shape[3] in (128, 256) and int(radius) in (3, 4)
                and 1 <= len(f2s) <= 4 and os.environ.get("UFR_ALTCORR_PLANES", "1") != "0")

    def __init__(self, f1: torch.Tensor, f2s):
        self.B, self.H1, self.W1, self.C = (int(v) for v in f1.shape)
        self.maps = []
        with torch.cuda.device(f1.device):
            for f in (f1, *f2s):
                L.require_hip(f, "feature map")
                if f.dim() != 4 or f.shape[0] != self.B or f.shape[3] != self.C or f.dtype != torch.float32:
                    raise RuntimeError("alt_corr planes: every map must be [B,H,W,C] float32 with fmap1's batch and channels")
                npix = f.shape[0] * f.shape[1] * f.shape[2]
                planes = torch.empty(3, npix * self.C, dtype=torch.bfloat16, device=f.device)
                L.check(L.lib().ufr_altcorr_planes_prepare(L.ptr(f), L.ptr(planes), planes.stride(0), npix, self.C, L.stream()),
                        "alt_corr planes prepare")
                self.maps.append((planes, int(f.shape[1]), int(f.shape[2])))
        self.levels = L.AltCorrPlaneLevels()
        self.levels.num_levels = len(f2s)
        for i, (planes, h2, w2) in enumerate(self.maps[1:]):
            self.levels.planes[i], self.levels.plane_stride[i] = planes.data_ptr(), planes.stride(0)
            self.levels.H2[i], self.levels.W2[i] = h2, w2
            self.levels.coord_scale[i] = 1.0 / 2 ** i                # corr.py:126

    def forward(self, coords: torch.Tensor, radius: int, scale: float, out: torch.Tensor | None = None) -> torch.Tensor:
        """coords [B, 2, H1, W1] -> [B, L (2r+1)^2, H1, W1] = scale * the stacked per-level windows."""
        rd = 2 * int(radius) + 1
        if out is None:
            out = torch.empty((self.B, self.levels.num_levels * rd * rd, self.H1, self.W1), dtype=torch.float32, device=coords.device)
        f1p = self.maps[0][0]
        L.check(L.lib().ufr_altcorr_planes_forward(L.ptr(f1p), f1p.stride(0), C.byref(self.levels), L.ptr(coords), L.ptr(out), self.B, self.H1,
                                                   self.W1, self.C, int(radius), float(scale), L.stream()), "alt_corr planes forward")
        return out


class AltCorrPyramidFunction(torch.autograd.Function):
    """All levels of one AlternateCorrBlock lookup (corr.py:121-137) as ONE launch of the matrix-core kernel
    (csrc/raft_altcorr_mfma.hip): out [B, L*(2r+1)^2, H, W] = stack_l alt_corr(fmap1, fmap2_l, coords / 2^l) / sqrt(dim).
    `shared` (a `_SharedGrad`): every lookup of one block (RAFT's 12 iterations) adds its adjoint into ONE set of
    buffers inside the kernels; the last adjoint to run hands them to autograd."""

    @staticmethod
    def forward(ctx, fmap1, coords, radius, scale, shared, *f2s):
        L.require_hip(fmap1, "fmap1")
        L.require_hip(coords, "coords", contiguous=False)
        coords = coords.contiguous()
        B, H1, W1, Cc = fmap1.shape
        if tuple(coords.shape) != (B, 2, H1, W1) or coords.dtype != torch.float32 or fmap1.dtype != torch.float32:
            raise RuntimeError("alt_corr pyramid: fmap1 [B,H,W,C] and coords [B,2,H,W], float32")
        for f in f2s:
            L.require_hip(f, "fmap2 level")
            if f.dim() != 4 or f.shape[0] != B or f.shape[3] != Cc or f.dtype != torch.float32:
                raise RuntimeError("alt_corr pyramid: every fmap2 level must be [B,H2,W2,C] float32")
        rd = 2 * int(radius) + 1
        with torch.cuda.device(fmap1.device):
            out = torch.empty((B, len(f2s) * rd * rd, H1, W1), dtype=torch.float32, device=fmap1.device)
            lv = _levels_struct(f2s)
            L.check(L.lib().ufr_altcorr_pyramid_forward(L.ptr(fmap1), C.byref(lv), L.ptr(coords), L.ptr(out), B, H1, W1, Cc,
                                                        int(radius), float(scale), L.stream()), "alt_corr pyramid forward")
        ctx.save_for_backward(fmap1, coords, *f2s)
        ctx.meta = (int(radius), float(scale))
        ctx.shared = shared
        if shared is not None:
            shared.pending += 1
        return out

    @staticmethod
    def backward(ctx, grad_out):
        fmap1, coords, *f2s = ctx.saved_tensors
        radius, scale = ctx.meta
        grad_out = grad_out.contiguous()
        B, H1, W1, Cc = fmap1.shape
        shared = ctx.shared
        with torch.cuda.device(fmap1.device):
            if shared is None or shared.acc is None:
                grads = [torch.empty_like(fmap1)] + [torch.empty_like(f) for f in f2s]
                ws = torch.empty(L.lib().ufr_altcorr_pyramid_workspace_bytes(B, H1, W1, Cc, radius, len(f2s)), dtype=torch.uint8,
                                 device=fmap1.device)
                accumulate = 0
                if shared is not None:
                    shared.acc = (grads, ws)
            else:
                grads, ws = shared.acc
                accumulate = 1
            lv = _levels_struct(f2s, grads[1:])
            L.check(L.lib().ufr_altcorr_pyramid_backward(L.ptr(fmap1), C.byref(lv), L.ptr(coords), L.ptr(grad_out), L.ptr(grads[0]),
                                                         L.ptr(ws), B, H1, W1, Cc, radius, scale, accumulate, L.stream()),
                    "alt_corr pyramid backward")
        if shared is not None:
            shared.pending -= 1
            if shared.pending > 0:          # more adjoints to come: hand autograd nothing yet
                return (None,) * (5 + len(f2s))
            shared.acc = None               # the last adjoint delivers the accumulated gradients
        return (grads[0], None, None, None, None, *grads[1:])     # coords are detached every RAFT iteration (raft.py:190)


class AlternateCorrBlock:
    """models/raft/corr.py:109-137, differentiable here (the reference calls the raw forward).  Feature maps with 128 or
    256 channels and radius 3 / 4 (both RAFT variants) take the one-launch matrix-core form; anything else the per-level
    drop-in calls."""

    def __init__(self, fmap1, fmap2, num_levels=4, radius=4, share_grad=False):
        """`share_grad=True` is RAFT's own opt-in (flownets/raft.py): every lookup of its loop reaches the loss and coords are
        detached (raft.py:190), so the lookups' feature-map adjoints accumulate in one buffer and the last adjoint delivers the
        sum.  A public caller gets one independent autograd node per lookup (the default)."""
        self.num_levels, self.radius = num_levels, radius
        self._maps, self._pyramid = (fmap1, fmap2), None
        # NHWC copies made once, not once per lookup as in the reference (:128-129).  The reference pools BOTH maps num_levels times
        # (:97-105) and reads fmap1's level 0 and fmap2's levels 0 .. num_levels - 1 only: those are what one launch per map makes
        # (raft_glue.fmap_pyramid, bit for bit the torch operators below it replaces).
        from ..raft_glue import fmap_pyramid
        p1 = fmap_pyramid(fmap1, 1)
        p2 = fmap_pyramid(fmap2, num_levels) if p1 is not None else None
        if p2 is not None:
            self._f1, self._f2 = p1[0], list(p2)
        else:
            self._f1 = self.pyramid[0][0].permute(0, 2, 3, 1).contiguous()
            self._f2 = [self.pyramid[i][1].permute(0, 2, 3, 1).contiguous() for i in range(num_levels)]
        self._shared = _SharedGrad() if share_grad else None
        dim = fmap1.shape[1]
        self._fused = (self._f1.is_cuda and self._f1.dtype == torch.float32 and dim in (128, 256) and radius in (3, 4)
                       and 1 <= num_levels <= 4)

    @property
    def pyramid(self):
        """The reference's list of (fmap1, fmap2) levels (corr.py:97-105), built when somebody asks for it."""
        if self._pyramid is None:
            fmap1, fmap2 = self._maps
            self._pyramid = [(fmap1, fmap2)]
            for _ in range(self.num_levels):
                fmap1 = F.avg_pool2d(fmap1, 2, stride=2)
                fmap2 = F.avg_pool2d(fmap2, 2, stride=2)
                self._pyramid.append((fmap1, fmap2))
        return self._pyramid

    def __call__(self, coords):
        dim = self._maps[0].shape[1]
        if self._fused and not (torch.is_grad_enabled() and coords.requires_grad):
            # (a caller that wants d / d coords takes the per-level drop-in below, which computes it like the reference's op)
            needs_grad = torch.is_grad_enabled() and (self._f1.requires_grad or any(f.requires_grad for f in self._f2))
            if needs_grad and self._shared is not None and self._shared.pending == 0:
                self._shared.acc = None          # first lookup after a completed (or abandoned) backward: nothing carries over
            return AltCorrPyramidFunction.apply(self._f1, coords, self.radius, 1.0 / math.sqrt(dim),
                                                self._shared if needs_grad else None, *self._f2)
        coords = coords.permute(0, 2, 3, 1)
        B, H, W, _ = coords.shape
        corr_list = []
        for i in range(self.num_levels):
            coords_i = (coords / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
            corr_list.append(AltCorrFunction.apply(self._f1, self._f2[i], coords_i, self.radius).squeeze(1))
        corr = torch.stack(corr_list, dim=1).reshape(B, -1, H, W)
        return corr / math.sqrt(dim)

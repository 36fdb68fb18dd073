"""Native PWC-DC-Net head: everything of models/PWCNet.py:225-367 behind pyramid level 2 -- pyramid levels 3-6, the five
coarse-to-fine decoder stages (cost volume, DenseNet block, flow / feature upsampling) and the dilated context network --
forward AND data gradient as an explicit schedule of hand-written gfx950 kernels.  Config C4 of BASELINE.json.

    3x3 convolutions (stride 1 / 2, dilation 1..16)   csrc/igemm.hip: implicit GEMMs on bf16 split planes (six products)
    predict_flow*, dc_conv7, upfeat*                  csrc/engine_small.hip (per-pixel GEMM + gather on the matrix cores)
    deconv* (2 -> 2), warp, 9x9 cost volume           csrc/small_cout.hip, pwc_warp.hip, correlation*.hip (NCHW float32)

Layout.  A decoder stage's DenseNet block (PWCNet.py:78-113: x = cat((conv_i(x), x), 1) five times) is ONE plane buffer
D_k whose chunks are [conv_4 32 | conv_3 64 | conv_2 96 | conv_1 128 | conv_0 128 | x]: conv_i reads the chunk suffix
behind its own output, predict_flow / upfeat / dc_conv1 read all of it -- no concatenation is ever copied.  The stage
input x = cat(corr 81, c1, up_flow 2, up_feat 2) (PWCNet.py:287) is stored as [corr 81, up_flow 2, up_feat 2, 11 zeros |
c1]; every weight is re-indexed once for that order.

Backward.  The data gradient of a DenseNet block is a DenseNet block in reverse: the gradient of segment j is the sum over
the later convolutions i > j of W_i[:, segment j]^T * gz_i, i.e. ONE 3x3 convolution over the gradient planes of all later
outputs, which sit in one buffer gz_k in the same order.  Each segment is written once, by a GEMM epilogue that adds the
two-channel consumers' part (predict_flow^T, upfeat^T: a float32 sum) and applies LeakyReLU' -- no read-modify-write of
float32 sums, no separate activation-gradient passes.  Parameters are frozen (data gradients only).
"""
from __future__ import annotations

import ctypes as C

import os

import torch
from ._lib import engine_cache as _engine_cache

from . import _lib as L
from . import igemm as ig
from . import spatial_correlation_sampler_backend as correlation
from .flownetc_engine import _pack_flow_head, _pack_flow_head_mfma, _pack_flow_tail_mfma

SEG = {4: 32, 3: 64, 2: 96, 1: 128, 0: 128}          # output channels of conv{k}_i (PWCNet.py:80-112)
A_OFF = {4: 0, 3: 1, 2: 3, 1: 6, 0: 10}              # first chunk of conv_i's output inside D_k
X0 = 14                                              # first chunk of the stage input x
FEAT = {2: 32, 3: 64, 4: 96, 5: 128, 6: 196}         # pyramid channels (PWCNet.py:55-72)
FLOW_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}     # PWCNet.py:286, :301, :316, :332
NCORR = 81
DC = ((1, 128, 1), (2, 128, 2), (3, 128, 4), (4, 96, 8), (5, 64, 16), (6, 32, 1))      # dc_conv{i}: outputs, dilation


def _x_map(lvl: int):
    """Buffer position (relative to chunk X0) of every channel of the reference's stage input."""
    if lvl == 6:
        return list(range(NCORR))
    c = FEAT[lvl]
    return list(range(NCORR)) + [96 + j for j in range(c)] + [81, 82] + [83, 84]


def _x_chunks(lvl: int) -> int:
    return 3 if lvl == 6 else 3 + ig.pad32(FEAT[lvl]) // 32


def _remap_in(weight: torch.Tensor, cmap, cbuf: int) -> torch.Tensor:
    """Conv2d weight [N, Cref, k, k] -> [N, cbuf, k, k] with input channel c moved to cmap[c] (zeros elsewhere)."""
    w = torch.zeros(weight.shape[0], cbuf, *weight.shape[2:], dtype=torch.float32, device=weight.device)
    w[:, torch.tensor(cmap, device=weight.device)] = weight.detach().float()
    return w


def _pack_tail_bwd(weight: torch.Tensor) -> torch.Tensor:
    """ConvTranspose2d(C, 2, 4, 2, 1).weight [C,2,4,4] (C in buffer order, a multiple of 32) -> [chunks][16][2][32] float32
    (csrc/engine_small.hip `flow_tail_planes_bwd`)."""
    c = weight.shape[0]
    return weight.detach().float().reshape(c // 32, 32, 2, 16).permute(0, 3, 2, 1).contiguous()


class _PyramidPrefix:
    """Pyramid levels 1-2 (PWCNet.py:55-60, :235-240: six 3x3 convolutions, strides 2,1,1,2,1,1, 3 -> 16 -> 32 channels) of `n`
    raw frames of h x w pixels on the igemm, forward and (optionally) data gradient.  The RGB -> BGR flip of PWCNet.py:230-231
    is folded into conv1a's weights.  Used for the full frames once per attack() call and for the attack's window per iteration."""

    def __init__(self, eng, n: int, h: int, w: int, backward: bool, f2_out: ig.Planes | None = None):
        net, dev = eng.net, eng.dev
        self.n, self.h, self.w = n, h, w
        dims = {0: (h, w), 1: (h // 2, w // 2), 2: (h // 4, w // 4)}
        P = lambda lvl, chunks=1: ig.Planes(n, dims[lvl][0], dims[lvl][1], chunks, dev)
        self.x, self.a1, self.aa1, self.b1, self.a2, self.aa2 = P(0), P(1), P(1), P(1), P(2), P(2)
        self.f2 = f2_out if f2_out is not None else P(2)             # the full-frame prefix writes the head's own input planes
        names = [nm for _, _, first, second, third in net._PYRAMID[:2] for nm in ("conv" + first, "conv" + second, "conv" + third)]
        acts = [self.x, self.a1, self.aa1, self.b1, self.a2, self.aa2, self.f2]
        lv = [0, 1, 1, 1, 2, 2, 2]
        plans = []
        for i, nm in enumerate(names):
            conv = getattr(net, nm)[0]
            wgt = conv.weight.detach().flip(1) if i == 0 else conv.weight.detach()
            stride = 2 if i in (0, 3) else 1
            plans.append(("fwd", i, ig.conv_forward_weights(wgt, stride, 1), acts[i], dims[lv[i + 1]], dims[lv[i + 1]],
                          dict(out_planes=acts[i + 1], bias=conv.bias.detach().float().contiguous())))
        if backward:
            self.gz = [None] + [P(lv[i]) for i in range(1, 7)]          # gradient planes of the six outputs
            self.G_x = ig.GradSum(n, h, w, 1, dev)
            self.f2_nchw = torch.zeros(n, FEAT[2], *dims[2], dtype=torch.float32, device=dev)
            self.gx = torch.zeros(n, 3, h, w, dtype=torch.float32, device=dev)
            for i in range(5, -1, -1):
                conv = getattr(net, names[i])[0]
                wgt = conv.weight.detach().flip(1) if i == 0 else conv.weight.detach()
                stride = 2 if i in (0, 3) else 1
                kw = dict(out_f32=self.G_x) if i == 0 else dict(out_planes=self.gz[i], mask=acts[i])
                plans.append(("bwd", i, ig.conv_backward_weights(wgt, stride, 1), self.gz[i + 1], dims[lv[i + 1]], dims[lv[i]], kw))
        conv1a = getattr(net, names[0])[0]
        self._w1a = conv1a.weight.detach().flip(1).float().contiguous()              # [16, 3 (BGR order folded), 3, 3]
        self._b1a = conv1a.bias.detach().float().contiguous()
        direct = os.environ.get("UFR_PWC_PYRAMID_DIRECT", "1") != "0"
        self._direct1a = direct and h % 2 == 0 and w % 2 == 0 and self._w1a.shape[0] <= 32
        # conv1aa / conv1b (16 -> 16 at half resolution: 12 % of their implicit GEMM is real work): [16 c][9 taps][16 o] weights
        self._direct16 = {}
        for i in (1, 2):
            c16 = getattr(net, names[i])[0]
            if direct and tuple(c16.weight.shape) == (16, 16, 3, 3):
                self._direct16[i] = (c16.weight.detach().float().permute(1, 2, 3, 0).reshape(16, 9, 16).contiguous(),
                                     c16.bias.detach().float().contiguous())
        sized = []
        for kind, i, wi, x, rows, out_hw, kw in plans:
            kw["variant"] = eng._variant_for(wi)
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            bm, target = eng._tile(wi, kw["variant"])
            sized.append(ig.splitk_for(n * rows[0] * rows[1], wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target,
                                       min_ktiles=4))
        need = max([len(p[2].phases) * S * n * p[4][0] * p[4][1] * p[2].Npad for p, S in zip(plans, sized) if S > 1] + [1])
        self.ws = torch.empty(need, dtype=torch.float32, device=dev)
        self.fwd, self.bwd, self.wi = {}, {}, {}
        for (kind, i, wi, x, rows, out_hw, kw), S in zip(plans, sized):
            launch = ig.make_launch(wi, x, 0, rows, out_hw, splitk=S, ws=self.ws if S > 1 else None, **kw)
            (self.fwd if kind == "fwd" else self.bwd)[i] = launch
            self.wi[(kind, i)] = wi

    @torch.no_grad()
    def forward(self, frames: torch.Tensor) -> ig.Planes:
        """Level-2 features (planes) of the frame stack [n, 3, h, w]."""
        if self._direct1a:            # conv1a straight from the raw frames (3 of 32 K channels and 16 of 64 columns are real on the igemm)
            L.check(L.lib().ufr_conv3x3s2_c3_planes(L.ptr(frames.contiguous()), L.ptr(self._w1a), L.ptr(self._b1a), ig.LEAKY, L.ptr(self.a1.t),
                                                    self.a1.plane_stride, 0, self.n, self._w1a.shape[0], self.h, self.w, L.stream()),
                    "conv1a (direct)")
        else:
            self.x.load_nchw(frames.contiguous(), 0)
            self.fwd[0]()
        acts = (self.x, self.a1, self.aa1, self.b1)
        for i in range(1, 6):
            if i in self._direct16:
                wt, b = self._direct16[i]
                src, dst = acts[i], acts[i + 1]
                L.check(L.lib().ufr_conv3x3_c16_planes(L.ptr(src.t), src.plane_stride, 0, L.ptr(wt), L.ptr(b), ig.LEAKY, L.ptr(dst.t),
                                                       dst.plane_stride, 0, self.n, src.H, src.W, L.stream()), "conv 16 -> 16 (direct)")
            else:
                self.fwd[i]()
        return self.f2

    @torch.no_grad()
    def backward(self, g_f2: torch.Tensor) -> torch.Tensor:
        """d loss / d frames [n, 3, h, w] from d loss / d (level-2 features) [n, 32, h/4, w/4] (NCHW)."""
        gz = self.gz[6]
        L.check(L.lib().ufr_nchw_grad_to_planes(L.ptr(g_f2), L.ptr(self.f2_nchw), L.ptr(gz.t), gz.plane_stride, 0, self.n, FEAT[2],
                                                self.h // 4, self.w // 4, ig.LEAKY, L.stream()), "level-2 gradient -> planes")
        for i in range(5, -1, -1):
            self.bwd[i]()
        return self.G_x.to_nchw(3, 0, slope=1.0, out=self.gx)


class PwcHeadEngine:
    def __init__(self, net, B: int, H: int, W: int, device):
        if H % 64 or W % 64:
            raise ValueError("PWC-Net head engine: frame sides must be multiples of 64")
        L.lib()
        self.net, self.B, self.H, self.W, self.dev = net, int(B), int(H), int(W), torch.device(device)
        self.grid = {k: (H >> k, W >> k) for k in range(2, 7)}
        self.generation = 0
        self._pipe_variant = 6
        self._narrow_variant = 7             # (same call, c4: 17.52 ms against 17.71 with the single-stage 128 x 64 tile)
        self._build()

    # ------------------------------------------------------------------------------------------------ set-up
    def _conv(self, name):
        return getattr(self.net, name)[0]

    def _variant_for(self, wi):
        """128-column launches: the ping-pong kernel.  64-column launches (64 / 32 outputs behind up to 565 input channels):
        its tap-reuse form with 256 x 64 tiles -- the activation rows are 80 % of their L2 -> LDS bytes and a horizontal run
        of taps is staged once (csrc/igemm.hip variant 7; launches it does not cover run the single-stage 128 x 64 tile)."""
        return self._pipe_variant if wi.Npad % 128 == 0 else self._narrow_variant

    def _tile(self, wi, variant):
        if wi.Npad % 128:
            return (256, 256) if variant == 7 else (128, 768)
        return {4: (64, 1024), 5: (128, 512), 6: (256, 256), 7: (256, 256)}.get(variant, (128, 768))

    def _build(self):
        net, B, dev, g = self.net, self.B, self.dev, self.grid
        f32 = dict(dtype=torch.float32, device=dev)
        P = lambda n, k, chunks: ig.Planes(n, g[k][0], g[k][1], chunks, dev)
        G = lambda n, k, chunks: ig.GradSum(n, g[k][0], g[k][1], chunks, dev)
        Z = lambda n, c, k: torch.zeros(n, c, *g[k], **f32)
        plans, self.fwd, self.bwd = [], {}, {}

        def plan(key, kind, wi, x, in_chunk0, rows_k, out_k, **kw):
            """Queue a launch over the row grid of level rows_k whose output lives on level out_k."""
            n = x.B
            kw.setdefault("variant", self._variant_for(wi))
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            bm, target = self._tile(wi, kw["variant"])
            S = ig.splitk_for(n * g[rows_k][0] * g[rows_k][1], wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target,
                              min_ktiles=4)
            kw["variant"], S = ig.tuned(wi, n * g[rows_k][0] * g[rows_k][1], kw, kw["variant"], S, rows=g[rows_k])
            plans.append((key, kind, wi, x, in_chunk0, g[rows_k], g[out_k], S, kw))

        bias = lambda name: self._conv(name).bias.detach().float().contiguous()
        # ---- pyramid levels 3-6 on both frames (2B images) ---------------------------------------------------------------
        self.F = {2: P(2 * B, 2, 1)}                                   # feature planes; level 2 is the engine's input
        self.F_nchw = {k: Z(2 * B, FEAT[k], k) for k in range(3, 7)}
        self.Fa, self.Faa, self.gzF, self.gzFa, self.gzFaa, self.T = {}, {}, {}, {}, {}, {}
        self.G_F = {k: Z(2 * B, FEAT[k], k) for k in range(2, 7)}      # d loss / d features, NCHW: [c1 half | c2 half]
        for k, (_, cout, first, second, third) in zip(range(3, 7), net._PYRAMID[2:]):
            ch = ig.pad32(cout) // 32
            self.Fa[k], self.Faa[k], self.F[k] = P(2 * B, k, ch), P(2 * B, k, ch), P(2 * B, k, ch)
            self.gzF[k], self.gzFaa[k], self.gzFa[k] = P(2 * B, k, ch), P(2 * B, k, ch), P(2 * B, k, ch)
            self.T[k - 1] = G(2 * B, k - 1, self.F[k - 1].chunks)      # conv{k}a's data gradient = part of d / d F[k-1]
            names = ("conv" + first, "conv" + second, "conv" + third)
            w = [self._conv(n).weight for n in names]
            plan(("pyr", k, 0), "fwd", ig.conv_forward_weights(w[0], 2, 1), self.F[k - 1], 0, k, k, out_planes=self.Fa[k], bias=bias(names[0]))
            plan(("pyr", k, 1), "fwd", ig.conv_forward_weights(w[1], 1, 1), self.Fa[k], 0, k, k, out_planes=self.Faa[k], bias=bias(names[1]))
            plan(("pyr", k, 2), "fwd", ig.conv_forward_weights(w[2], 1, 1), self.Faa[k], 0, k, k, out_planes=self.F[k], bias=bias(names[2]))
            plan(("pyr", k, 2), "bwd", ig.conv_backward_weights(w[2], 1, 1), self.gzF[k], 0, k, k, mask=self.Faa[k], out_planes=self.gzFaa[k])
            plan(("pyr", k, 1), "bwd", ig.conv_backward_weights(w[1], 1, 1), self.gzFaa[k], 0, k, k, mask=self.Fa[k], out_planes=self.gzFa[k])
            plan(("pyr", k, 0), "bwd", ig.conv_backward_weights(w[0], 2, 1), self.gzFa[k], 0, k, k - 1, out_f32=self.T[k - 1])
        # ---- decoder stages ------------------------------------------------------------------------------------------------
        self.D, self.gzD, self.G_D, self.G_x, self.P4 = {}, {}, {}, {}, {}
        self.x_nchw, self.gx_nchw, self.corr, self.g_corr = {}, {}, {}, {}
        self.gx_upflow, self.gx_c1 = {}, {}
        self.flow = {k: Z(B, 2, k) for k in range(2, 7)}
        self.g_flow = {k: Z(B, 2, k) for k in range(2, 7)}
        self.up_flow, self.up_flow_s, self.up_feat = {}, {}, {}
        self.warped, self.g_warped, self.g_c1corr = {}, {}, {}
        self.g_upflow, self.g_upfeat, self.g_flow_s, self.warp_ws = {}, {}, {}, {}
        self.pf_w, self.pf_wm, self.pf_b = {}, {}, {}
        self.uf_wm, self.uf_wb, self.uf_b = {}, {}, {}
        self.dec_w, self.dec_b = {}, {}
        for k in (6, 5, 4, 3, 2):
            xc = _x_chunks(k)
            nch, xw = X0 + xc, xc * 32
            gz0 = 4 if k == 2 else 0                                  # level 2: dc_conv1's gradient planes in front
            self.D[k], self.gzD[k] = P(B, k, nch), P(B, k, gz0 + X0)
            self.G_D[k], self.G_x[k] = G(B, k, nch), G(B, k, xc)
            cx = NCORR if k == 6 else NCORR + 4
            self.x_nchw[k], self.gx_nchw[k] = Z(B, cx, k), Z(B, xw, k)            # (level 6 only: its x is the cost volume alone)
            if k < 6:
                self.gx_upflow[k], self.gx_c1[k] = Z(B, 2, k), Z(B, FEAT[k], k)   # the stage input's gradient, member by member
            self.corr[k], self.g_corr[k] = Z(B, NCORR, k), Z(B, NCORR, k)
            self.g_c1corr[k], self.g_warped[k] = Z(B, FEAT[k], k), Z(B, FEAT[k], k)
            if k < 6:
                self.up_flow[k], self.up_flow_s[k], self.up_feat[k] = Z(B, 2, k), Z(B, 2, k), Z(B, 2, k)
                self.warped[k] = Z(B, FEAT[k], k)
                self.g_upflow[k], self.g_upfeat[k], self.g_flow_s[k] = Z(B, 2, k), Z(B, 2, k), Z(B, 2, k)
                # the warp adjoint's table of sampling boxes (owner-computes, csrc/pwc_warp.hip): static like every buffer of the engine
                nb = int(L.lib().ufr_pwc_warp_backward_workspace_bytes(B, *self.grid[k]))
                self.warp_ws[k] = (torch.empty((nb + 15) // 16 * 4, dtype=torch.int32, device=dev), nb)
            xmap = _x_map(k)
            wbuf = {}                                                  # conv{k}_i in buffer channel order
            for i in range(5):
                pre = (X0 - (A_OFF[i - 1] if i else X0)) * 32          # channels of conv_{i-1} .. conv_0 in front of x
                wref = self._conv(f"conv{k}_{i}").weight
                wbuf[i] = _remap_in(wref, list(range(pre)) + [pre + m for m in xmap], pre + xw)
            # DenseNet forward push: conv_2 has 96 outputs (a quarter of its 128-column tiles would be padding) and conv_4 has 32
            # behind the longest reduction of the block (half of a 64-column tile).  conv_4's sum over the chunks conv_2 reads
            # anyway ([conv_1 | conv_0 | x]) rides in conv_2's spare columns as a raw fp32 partial (`tail`); conv_4's own launch
            # then reduces over [conv_3 | conv_2] only and takes the partial back before its bias.
            self.P4[k] = G(B, k, 1)
            skip = (A_OFF[1] - A_OFF[3]) * 32                          # conv_4's input channels in front of conv_1: conv_3, conv_2
            for i in range(5):
                c0 = A_OFF[i - 1] if i else X0
                w, kw = wbuf[i], {}
                if i == 2:
                    w = torch.cat((wbuf[2], wbuf[4][:, skip:]), 0)     # [96 + 32, conv_1 | conv_0 | x, 3, 3]
                    b2 = torch.cat((bias(f"conv{k}_2"), torch.zeros(SEG[4], **f32)))
                    kw = dict(tail=self.P4[k], tail_n0=SEG[2], bias=b2)
                elif i == 4:
                    w = wbuf[4][:, :skip].contiguous()
                    kw = dict(add=self.P4[k], add_chunk0=0)
                kw.setdefault("bias", bias(f"conv{k}_{i}"))
                plan(("dec", k, i), "fwd", ig.conv_forward_weights(w, 1, 1), self.D[k], c0, k, k, out_planes=self.D[k],
                     out_chunk0=A_OFF[i], **kw)
            full_map = list(range(X0 * 32)) + [X0 * 32 + m for m in xmap]
            pf = getattr(net, f"predict_flow{k}")
            wpf = _remap_in(pf.weight, full_map, nch * 32)
            self.pf_w[k], self.pf_wm[k] = _pack_flow_head(wpf), _pack_flow_head_mfma(wpf)
            self.pf_b[k] = pf.bias.detach().float().contiguous()
            if k > 2:
                uf = getattr(net, f"upfeat{k}")                        # ConvTranspose2d(C, 2, 4, 2, 1): weight [C, 2, 4, 4]
                wuf = _remap_in(uf.weight.detach().permute(1, 0, 2, 3), full_map, nch * 32)        # [2, Cbuf, 4, 4]
                self.uf_wm[k] = _pack_flow_tail_mfma(wuf)
                self.uf_wb[k] = _pack_tail_bwd(wuf.permute(1, 0, 2, 3).contiguous())
                self.uf_b[k] = uf.bias.detach().float().contiguous()
                dec = getattr(net, f"deconv{k}")
                self.dec_w[k], self.dec_b[k] = dec.weight.detach().float().contiguous(), dec.bias.detach().float().contiguous()
            # the reversed DenseNet block: sources in gz order = [dc_conv1 (level 2) | conv_4 | conv_3 | ... ]
            src = []                                                   # (id, weight in buffer order, first buffer chunk it reads)
            if k == 2:
                self.w_dc1 = _remap_in(self._conv("dc_conv1").weight, full_map, nch * 32)
                src.append(("dc1", self.w_dc1, 0))
            for i in (4, 3, 2, 1, 0):
                src.append((i, wbuf[i], A_OFF[i - 1] if i else X0))
            for j in (4, 3, 2, 1, 0, "x"):
                dst0, width = (X0, xw) if j == "x" else (A_OFF[j], SEG[j])
                later = [(w, c0) for sid, w, c0 in src if sid == "dc1" or j == "x" or sid > j]      # the convolutions that read segment j
                if not later:
                    continue                                           # conv_4 of levels 3-6: predict_flow^T and upfeat^T only
                if k == 2 and j in (4, 3):
                    # level 2: conv_4 (32 channels) and conv_3 (64) both take dc_conv1's gradient.  ONE 128-column launch over
                    # dc_conv1's four gradient chunks computes conv_4's whole data gradient (columns 0-31, through the epilogue)
                    # and conv_3's share (columns 32-95: raw sums ADDED onto G_D's conv_3 chunks, the tail of csrc/igemm.hip);
                    # conv_3's own launch then reduces over conv_4's gradient chunk only.  (32- and 64-column launches over a
                    # 36-step reduction ran half-empty tiles: 0.49 ms -> 0.34 ms.)
                    if j == 4:
                        wv = self.w_dc1[:, :(SEG[4] + SEG[3])]         # [dc_conv1's 128 outputs, conv_4 | conv_3 channels, 3, 3]
                        wi = ig.conv_backward_weights(wv, 1, 1)
                        plan(("dec", 2, 4), "bwd", wi, self.gzD[2], 0, 2, 2, add=self.G_D[2], add_chunk0=A_OFF[4], mask=self.D[2],
                             mask_chunk0=A_OFF[4], out_planes=self.gzD[2], out_chunk0=gz0 + A_OFF[4], tail=self.G_D[2], tail_n0=SEG[4],
                             tail_chunk0=A_OFF[3], tail_accumulate=True)
                    else:
                        wv = wbuf[4][:, :SEG[3]]                       # conv_4's weights on conv_3's channels (conv_4 reads from chunk 1)
                        wi = ig.conv_backward_weights(wv, 1, 1)
                        plan(("dec", 2, 3), "bwd", wi, self.gzD[2], gz0 + A_OFF[4], 2, 2, add=self.G_D[2], add_chunk0=A_OFF[3],
                             mask=self.D[2], mask_chunk0=A_OFF[3], out_planes=self.gzD[2], out_chunk0=gz0 + A_OFF[3])
                    continue
                # stacked like the gradient planes in gzD: [gz channels of the later outputs, segment channels, 3, 3]
                wv = torch.cat([w[:, (dst0 - c0) * 32:(dst0 - c0) * 32 + width] for w, c0 in later], 0)
                wi = ig.conv_backward_weights(wv, 1, 1)
                if wi.KC != gz0 + (X0 if j == "x" else A_OFF[j]):
                    raise AssertionError("reversed DenseNet block: gradient planes and weights disagree")
                if j == "x":
                    plan(("dec", k, "x"), "bwd", wi, self.gzD[k], 0, k, k, add=self.G_D[k], add_chunk0=X0, out_f32=self.G_x[k])
                else:
                    plan(("dec", k, j), "bwd", wi, self.gzD[k], 0, k, k, add=self.G_D[k], add_chunk0=A_OFF[j], mask=self.D[k],
                         mask_chunk0=A_OFF[j], out_planes=self.gzD[k], out_chunk0=gz0 + A_OFF[j])
        # ---- context network (PWCNet.py:145-152, :340-346) ----------------------------------------------------------------
        self.dc, self.gz_dc = {}, {}
        prev, prev_c0 = self.D[2], 0
        for i, cout, dil in DC:
            self.dc[i] = P(B, 2, ig.pad32(cout) // 32)
            w = self.w_dc1 if i == 1 else self._conv(f"dc_conv{i}").weight
            plan(("dc", i), "fwd", ig.conv_forward_weights(w, 1, dil, dil), prev, prev_c0, 2, 2, out_planes=self.dc[i], bias=bias(f"dc_conv{i}"))
            prev = self.dc[i]
        for i, cout, dil in DC[1:]:                                   # dc_conv1's data gradient is part of the reversed block
            self.gz_dc[i] = P(B, 2, ig.pad32(cout) // 32)
        for i, cout, dil in reversed(DC[1:]):
            out = self.gzD[2] if i == 2 else self.gz_dc[i - 1]
            plan(("dc", i), "bwd", ig.conv_backward_weights(self._conv(f"dc_conv{i}").weight, 1, dil, dil), self.gz_dc[i], 0, 2, 2,
                 mask=self.dc[i - 1], out_planes=out)
        self.G_dc6 = G(B, 2, 1)
        w7 = self.net.dc_conv7.weight
        self.dc7_w, self.dc7_wm = _pack_flow_head(w7), _pack_flow_head_mfma(w7)
        self.dc7_b = self.net.dc_conv7.bias.detach().float().contiguous()
        self.dc7_out, self.flow2 = Z(B, 2, 2), Z(B, 2, 2)
        self.g_f2 = Z(2 * B, FEAT[2], 2)
        self.f2_cache = Z(2 * B, FEAT[2], 2)                            # the attack step's cached level-2 features (NCHW)
        self._prefix, self._wprefixes, self._wprefix = None, {}, None
        self.flow_out, self.flow_scale = self.flow2, 20.0             # what the step's fused loss kernel reads (PWCNet.py:367)
        # ---- one split-K workspace for every launch ------------------------------------------------------------------------
        need = max([len(wi.phases) * S * x.B * rows[0] * rows[1] * wi.Npad for _, _, wi, x, _, rows, _, S, _ in plans if S > 1] + [1])
        self.ws = torch.empty(need, **f32)
        self._plans = {}
        for key, kind, wi, x, c0, rows, out_hw, S, kw in plans:
            launch = ig.make_launch(wi, x, c0, rows, out_hw, splitk=S, ws=self.ws if S > 1 else None, **kw)
            (self.fwd if kind == "fwd" else self.bwd)[key] = launch
            self._plans[(kind, key)] = wi
        self._corr_p = correlation._params(1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)              # PWCNet.py:42-50

    def launch_table(self):
        """Every prepared igemm launch with its algorithmic work: [(name, 'fwd' | 'bwd', launch, GFLOP)]."""
        rows = []
        for kind, table in (("fwd", self.fwd), ("bwd", self.bwd)):
            for key, launch in table.items():
                d = launch.desc
                rows.append(("_".join(str(v) for v in key), kind, launch, self._plans[(kind, key)].flops(d.B * d.Hr * d.Wr) / 1e9))
        return rows

    # ------------------------------------------------------------------------------------------------ small launches
    def _pf_forward(self, src, chunks, wm, b, out):
        L.check(L.lib().ufr_flow_head_planes_forward_mfma(L.ptr(src.t), src.plane_stride, 0, chunks, L.ptr(wm), wm.shape[0], L.ptr(b), L.ptr(out),
                                                          src.B, src.H, src.W, L.stream()), "predict_flow forward (mfma)")

    def _pf_backward(self, gy, w, Gs, chunks, accumulate=False):
        L.check(L.lib().ufr_flow_head_planes_backward(L.ptr(gy), L.ptr(w), w.shape[0], L.ptr(Gs.t), Gs.chunks, 0, chunks, Gs.B, Gs.H, Gs.W, int(accumulate),
                                                      L.stream()), "predict_flow backward")

    def _finalize(self, Gs, g_chunk0, mask, mask_chunk0, out, out_chunk0, chunks):
        L.check(L.lib().ufr_grad_finalize(L.ptr(Gs.t), g_chunk0, L.ptr(mask.t), mask_chunk0, L.ptr(out.t), out.plane_stride,
                                          out_chunk0, Gs.M, chunks, ig.LEAKY, L.stream()), "gradient finalize")

    def _cat_call(self, fn, members, k, *front, after=()):
        """One of the two stage-input layout calls (include/ufr_hip.h: ufr_nchw_cat_to_planes / ufr_chunks_to_nchw_cat) over the
        members (corr 81, up_flow 2, up_feat 2, c1) of level k's x at buffer channels 0, 81, 83, 96."""
        ptrs = (C.c_void_p * 4)(*[m.data_ptr() for m in members])
        chans = (C.c_int * 4)(NCORR, 2, 2, FEAT[k])
        first = (C.c_int * 4)(0, NCORR, NCORR + 2, 96)
        if after:          # backward: (G, chunk0, chunks, dsts, channels, first, n, act0, pos, neg, B, H, W, stream)
            L.check(fn(*front, ptrs, chans, first, 4, *after), "stage input backward")
        else:              # forward: (srcs, channels, first, n, planes, plane_stride, chunk0, chunks, B, H, W, stream)
            L.check(fn(ptrs, chans, first, 4, *front), "stage input forward")

    def _corr_forward(self, a, b, out):
        n, c, h, w = a.shape
        L.check(L.lib().ufr_corr_forward_fused(L.ptr(a), L.ptr(b), L.ptr(out), L.UFR_F32, n, c, h, w, C.byref(self._corr_p),
                                               1.0 / c, ig.LEAKY, L.stream()), "correlation forward")

    def _corr_backward(self, a, b, g, ga, gb):
        n, c, h, w = a.shape
        L.check(L.lib().ufr_corr_backward(L.ptr(a), L.ptr(b), L.ptr(g), L.ptr(ga), L.ptr(gb), L.UFR_F32, n, c, h, w,
                                          C.byref(self._corr_p), L.stream()), "correlation backward")

    # ------------------------------------------------------------------------------------------------ the schedule
    # ---- the attack step's cached prefix (patch_attack.py): full-frame levels 1-2 once per call, the window per iteration
    @torch.no_grad()
    def prefix_full(self, frames_a: torch.Tensor, frames_b: torch.Tensor):
        """New frames: pyramid levels 1-2 of both full frames straight into the head's input planes (+ NCHW for the cost
        volume / warp kernels of level 2)."""
        if self._prefix is None:
            self._prefix = _PyramidPrefix(self, 2 * self.B, self.H, self.W, backward=False, f2_out=self.F[2])
        self._prefix.forward(torch.cat((frames_a, frames_b), 0))
        self.F[2].to_nchw(FEAT[2], 0, out=self.f2_cache)
        self.F_nchw[2] = self.f2_cache

    @torch.no_grad()
    def load_prefix_features(self, f2: torch.Tensor):
        """The same from level-2 features the caller already holds (train()'s clean forward seeds the attack's cache)."""
        self.f2_cache.copy_(f2[:2 * self.B])
        self.F_nchw[2] = self.f2_cache
        self.F[2].load_nchw(self.f2_cache, 0)

    def window_prefix(self, wh: int, ww: int) -> _PyramidPrefix:
        """One state per window size, at most four cached (least recently used goes); a step that captured graphs over a
        state holds the state itself (PatchAttackStep._wp_hold): captured graphs hold raw pointers into it."""
        P = self._wprefixes.pop((int(wh), int(ww)), None)
        if P is None:
            P = _PyramidPrefix(self, 2 * self.B, int(wh), int(ww), backward=True)
        self._wprefixes[(int(wh), int(ww))] = P
        while len(self._wprefixes) > 4:
            del self._wprefixes[next(iter(self._wprefixes))]
        self._wprefix = P
        return P

    @torch.no_grad()
    def window_prefix_forward(self, xw: torch.Tensor, win: torch.Tensor, margin: int):
        """Levels 1-2 of the window stack `xw` [2B, 3, wh, ww] (first frames, then second frames), patched into the cached
        full-frame level-2 features; the inexact rim (`margin` cells, cone.py) is skipped by the scatter."""
        wh, ww = int(xw.shape[2]), int(xw.shape[3])
        P = self.window_prefix(wh, ww)
        P.forward(xw.detach())
        P.f2.to_nchw(FEAT[2], 0, out=P.f2_nchw)
        lib, B, st = L.lib(), self.B, L.stream()
        h4, w4 = self.grid[2]
        L.check(lib.ufr_window_scatter_planes(L.ptr(P.f2_nchw), L.ptr(self.F[2].t), self.F[2].plane_stride, 0, L.ptr(win), B, 2 * B,
                                              FEAT[2], h4, w4, wh // 4, ww // 4, 4, int(margin), st), "window -> planes (level 2)")
        L.check(lib.ufr_window_scatter(L.ptr(P.f2_nchw), L.ptr(self.f2_cache), L.ptr(win), B, 2 * B, FEAT[2], h4, w4, wh // 4, ww // 4, 4,
                                       int(margin), st), "window -> nchw (level 2)")

    @torch.no_grad()
    def window_prefix_backward(self, gw: torch.Tensor) -> torch.Tensor:
        """d loss / d xw [2B, 3, wh, ww] from the window's level-2 gradient [2B, 32, wh/4, ww/4] (rim zeroed)."""
        return self._wprefix.backward(gw)

    @torch.no_grad()
    def forward_cached(self) -> torch.Tensor:
        """`forward` on the level-2 features the engine already holds (prefix_full / load_prefix_features + the window)."""
        self.generation += 1
        self.F_nchw[2] = self.f2_cache
        return self._forward_head()

    @torch.no_grad()
    def forward(self, f2: torch.Tensor) -> torch.Tensor:
        """Level-2 features of both frames [2B, 32, H/4, W/4] (first frames, then second frames; NCHW float32) -> flow2
        [B, 2, H/4, W/4] (PWCNet.py:355 before the x4 upsampling and the factor 20)."""
        L.require_hip(f2, "f2")
        B = self.B
        if tuple(f2.shape) != (2 * B, FEAT[2], *self.grid[2]) or f2.dtype != torch.float32:
            raise RuntimeError("PWC-Net head engine: level-2 features of another shape")
        self.generation += 1
        self.F_nchw[2] = f2
        self.F[2].load_nchw(f2, 0)
        return self._forward_head()

    @torch.no_grad()
    def _forward_head(self) -> torch.Tensor:
        B, lib, st = self.B, L.lib(), L.stream
        for k in range(3, 7):
            for i in range(3):
                self.fwd[("pyr", k, i)]()
            self.F[k].to_nchw(FEAT[k], 0, out=self.F_nchw[k])
        for k in (6, 5, 4, 3, 2):
            c1, c2 = self.F_nchw[k][:B], self.F_nchw[k][B:]
            D = self.D[k]
            if k == 6:
                self._corr_forward(c1, c2, self.x_nchw[6])
                D.load_nchw(self.x_nchw[6], X0)
            else:
                h, w = self.grid[k + 1]
                L.check(lib.ufr_deconv4x4s2_c2_forward(L.ptr(self.flow[k + 1]), L.ptr(self.dec_w[k + 1]), L.ptr(self.dec_b[k + 1]),
                                                       L.ptr(self.up_flow[k]), B, h, w, st()), "deconv forward")
                Dc = self.D[k + 1]
                L.check(lib.ufr_upfeat_planes_forward_mfma(L.ptr(Dc.t), Dc.plane_stride, 0, Dc.chunks, L.ptr(self.uf_wm[k + 1]),
                                                           self.uf_wm[k + 1].shape[0], L.ptr(self.uf_b[k + 1]), L.ptr(self.up_feat[k]), B, h, w, st()), "upfeat forward")
                torch.mul(self.up_flow[k], FLOW_SCALE[k], out=self.up_flow_s[k])
                L.check(lib.ufr_pwc_warp_forward(L.ptr(c2), L.ptr(self.up_flow_s[k]), L.ptr(self.warped[k]), B, FEAT[k], *self.grid[k], st()),
                        "warp forward")
                self._corr_forward(c1, self.warped[k], self.corr[k])
                # x = [corr 81 | up_flow 2 | up_feat 2 | 11 zeros | c1] straight into the planes (one pass instead of five)
                self._cat_call(lib.ufr_nchw_cat_to_planes, (self.corr[k], self.up_flow[k], self.up_feat[k], c1), k,
                               L.ptr(D.t), D.plane_stride, X0, _x_chunks(k), B, *self.grid[k], st())
            for i in range(5):
                self.fwd[("dec", k, i)]()
            self._pf_forward(D, D.chunks, self.pf_wm[k], self.pf_b[k], self.flow[k])
        for i, _, _ in DC:
            self.fwd[("dc", i)]()
        self._pf_forward(self.dc[6], 1, self.dc7_wm, self.dc7_b, self.dc7_out)
        torch.add(self.flow[2], self.dc7_out, out=self.flow2)
        return self.flow2

    @torch.no_grad()
    def backward(self, g_flow2: torch.Tensor) -> torch.Tensor:
        """d loss / d flow2 -> d loss / d (level-2 features) [2B, 32, H/4, W/4] (a static buffer)."""
        L.require_hip(g_flow2, "g_flow2")
        B, lib, st = self.B, L.lib(), L.stream
        # context network: flow2 = flow[2] + dc_conv7(...)
        self._pf_backward(g_flow2, self.dc7_w, self.G_dc6, 1)
        self._finalize(self.G_dc6, 0, self.dc[6], 0, self.gz_dc[6], 0, 1)
        for i in (6, 5, 4, 3, 2):
            self.bwd[("dc", i)]()                                      # dc_conv2's writes dc_conv1's gradient planes into gzD[2]
        g_flow = g_flow2
        for k in (2, 3, 4, 5, 6):
            D, Gd, gz = self.D[k], self.G_D[k], self.gzD[k]
            self._pf_backward(g_flow, self.pf_w[k], Gd, D.chunks)
            if k > 2:                                                  # + upfeat^T of the finer level's up_feat gradient
                L.check(lib.ufr_upfeat_planes_backward(L.ptr(self.g_upfeat[k - 1]), L.ptr(self.uf_wb[k]), self.uf_wb[k].shape[0], L.ptr(Gd.t), Gd.chunks, 0, D.chunks, B,
                                                       *self.grid[k], 1, st()), "upfeat backward")
                self._finalize(Gd, 0, D, 0, gz, 0, 1)                  # conv_4: no later convolution reads it
            for j in ((4, 3, 2, 1, 0) if k == 2 else (3, 2, 1, 0)):
                self.bwd[("dec", k, j)]()
            self.bwd[("dec", k, "x")]()
            c1, c2 = self.F_nchw[k][:B], self.F_nchw[k][B:]
            gF = self.G_F[k]
            # correlate = / C then LeakyReLU (PWCNet.py:42-50, :262): the sign of the activation is the pre-activation's
            if k == 6:
                gx = self.G_x[6].to_nchw(self.gx_nchw[6].shape[1], 0, slope=1.0, out=self.gx_nchw[6])
                torch.mul(gx[:, :NCORR], torch.where(self.x_nchw[6] > 0, 1.0 / FEAT[6], ig.LEAKY / FEAT[6]), out=self.g_corr[6])
                self._corr_backward(c1, c2, self.g_corr[6], gF[:B], gF[B:])
                break
            # d / d x, member by member, in one pass: the cost volume's part through its activation, up_flow, up_feat, c1
            self._cat_call(lib.ufr_chunks_to_nchw_cat, (self.g_corr[k], self.gx_upflow[k], self.g_upfeat[k], self.gx_c1[k]), k,
                           L.ptr(self.G_x[k].t), 0, _x_chunks(k), after=(L.ptr(self.corr[k]), 1.0 / FEAT[k], ig.LEAKY / FEAT[k], B,
                                                                          *self.grid[k], st()))
            self._corr_backward(c1, self.warped[k], self.g_corr[k], self.g_c1corr[k], self.g_warped[k])
            torch.add(self.gx_c1[k], self.g_c1corr[k], out=gF[:B])
            L.check(lib.ufr_pwc_warp_backward_owner(L.ptr(c2), L.ptr(self.up_flow_s[k]), L.ptr(self.g_warped[k]), L.ptr(gF[B:]),
                                                    L.ptr(self.g_flow_s[k]), L.ptr(self.warp_ws[k][0]), self.warp_ws[k][1], B, FEAT[k],
                                                    *self.grid[k], st()), "warp backward")
            torch.add(self.gx_upflow[k], self.g_flow_s[k], alpha=FLOW_SCALE[k], out=self.g_upflow[k])
            L.check(lib.ufr_deconv4x4s2_c2_backward_data(L.ptr(self.g_upflow[k]), L.ptr(self.dec_w[k + 1]), L.ptr(self.g_flow[k + 1]), B,
                                                         *self.grid[k + 1], st()), "deconv backward")
            g_flow = self.g_flow[k + 1]
        # pyramid levels 6..3: d / d F[k] = the decoder's part (NCHW) + the stride-2 convolution of the level above
        for k in (6, 5, 4, 3):
            gF = self.G_F[k]
            if k < 6:
                gF.add_(self.T[k].to_nchw(FEAT[k], 0, slope=1.0))
            gz = self.gzF[k]
            L.check(lib.ufr_nchw_grad_to_planes(L.ptr(gF), L.ptr(self.F_nchw[k]), L.ptr(gz.t), gz.plane_stride, 0, 2 * B, FEAT[k],
                                                *self.grid[k], ig.LEAKY, st()), "feature gradient -> planes")
            for i in (2, 1, 0):
                self.bwd[("pyr", k, i)]()
        torch.add(self.G_F[2], self.T[2].to_nchw(FEAT[2], 0, slope=1.0), out=self.g_f2)
        return self.g_f2


class _PwcEngineHead(torch.autograd.Function):
    """flow2 of PWC-Net's head as an autograd Function of the level-2 features (static buffers: see
    flownetc_engine._EngineHead for the generation rule)."""

    @staticmethod
    def forward(ctx, f2, engine):
        ctx.engine = engine
        out = engine.forward(f2.contiguous()).clone()
        ctx.generation = engine.generation
        return out

    @staticmethod
    def backward(ctx, g_flow2):
        if ctx.engine.generation != ctx.generation:
            raise RuntimeError("PWC-Net head engine: another forward of this network (same batch and frame size) ran before this "
                               "backward; its activations are gone.  Call backward() before the next forward, or set UFR_ENGINE=0")
        return ctx.engine.backward(g_flow2.contiguous()).clone(), None


def get_engine(net, B: int, H: int, W: int, device) -> PwcHeadEngine:
    """One engine per batch / frame size, cached on the module; rebuilt when the module's weights have changed since."""
    from .flownetc_engine import _weights_stamp
    key = (int(B), int(H), int(W), str(torch.device(device)))
    cache = _engine_cache(net, "_ufr_head_engines")
    stamp = _weights_stamp(net)
    eng = cache.get(key)
    if eng is None or eng.weights_stamp != stamp:
        eng = cache[key] = PwcHeadEngine(net, B, H, W, device)
        eng.weights_stamp = stamp
    return eng


def engine_head(net, f2a: torch.Tensor, f2b: torch.Tensor) -> torch.Tensor:
    B, _, h4, w4 = f2a.shape
    return _PwcEngineHead.apply(torch.cat((f2a, f2b), 0), get_engine(net, B, h4 * 4, w4 * 4, f2a.device))

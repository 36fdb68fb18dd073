"""Native FlowNetC head: everything of models/FlowNetC.py:121-197 behind conv1-3 -- correlation, conv_redir, conv3_1 ...
conv6_1, the coarse-to-fine refinement -- forward AND data gradient as an explicit schedule of hand-written gfx950
kernels, with no torch operator between the siamese features and flow2.

    convolutions / deconvolutions   csrc/igemm.hip: implicit GEMMs on bf16 split planes (float32-accurate, six products)
    predict_flow*, upsampled_flow*  csrc/engine_small.hip (HBM-bound passes over the concatenation buffers)
    correlation                     csrc/correlation*.hip (unchanged), cost volume converted once into conv3_1's input

Activations stay in the chunk-major plane layout between layers (igemm.Planes); a `torch.cat` of the reference
(FlowNetC.py:142, :167, :172, :177, :182) is a chunk offset into one buffer:

    in31 = [conv_redir 32 | corr 441]            15 chunks @ 1/8     cat3 = [conv3_1 256 | deconv3 128 | flow4_up 2]  13 chunks @ 1/8
    cat4 = [conv4_1 512 | deconv4 256 | up 2]    25 chunks @ 1/16    cat5 = [conv5_1 512 | deconv5 512 | up 2]        33 chunks @ 1/32
    cat2 = [conv2 128 | deconv2 64 | up 2]        7 chunks @ 1/4

The backward pass is the reverse schedule; a gradient with several consumers is summed in fp32 (igemm.GradSum) and the last
contributing GEMM applies LeakyReLU' and writes the gradient planes of the next GEMM directly (no separate passes where a
GEMM epilogue can do it).  Parameters are frozen: data gradients only (the reference's loss.backward() also computes
every weight gradient, main.py:573).  All buffers are allocated once; a step is a fixed sequence of launches, so the
attack's HIP graph captures it as is.
"""
from __future__ import annotations

import os

import torch
from ._lib import engine_cache as _engine_cache

from . import _lib as L
from . import igemm as ig

_HEADS = ((6, 1024), (5, 1026), (4, 770), (3, 386), (2, 194))


def _pack_flow_head(weight: torch.Tensor) -> torch.Tensor:
    """Conv2d(Cin, 2, 3, 1, 1).weight [2,Cin,3,3] -> [chunks][9][2][32] float32 (csrc/engine_small.hip)."""
    cin = weight.shape[1]
    chunks = ig.pad32(cin) // 32
    w = torch.zeros(2, chunks * 32, 9, dtype=torch.float32, device=weight.device)
    w[:, :cin] = weight.detach().float().reshape(2, cin, 9)
    return w.view(2, chunks, 32, 9).permute(1, 3, 0, 2).contiguous()


def _pack_flow_head_mfma(weight: torch.Tensor) -> torch.Tensor:
    """Conv2d(Cin, 2, 3, 1, 1).weight -> bf16 [chunks][3 planes][2][16][32]: plane p of w[o][32 ch + c][k] at n = 2 k + o
    (the B operand of csrc/engine_small.hip `flow_head_planes_fwd_mfma`; columns 18..31 are zero)."""
    cin = weight.shape[1]
    chunks = ig.pad32(cin) // 32
    w = torch.zeros(chunks * 32, 32, dtype=torch.float32, device=weight.device)          # [c][n]
    w[:cin, :18] = weight.detach().float().reshape(2, cin, 9).permute(1, 2, 0).reshape(cin, 18)
    wn = w.view(chunks, 32, 32).permute(0, 2, 1).contiguous()                              # [chunk][n][c]
    planes = ig._split3(wn).view(3, chunks, 32, 32).permute(1, 0, 2, 3).contiguous()      # [chunk][plane][n][c]
    return planes


def _pack_flow_tail_mfma(weight2: torch.Tensor) -> torch.Tensor:
    """The two flow rows of a ConvTranspose2d(Cin, Cout, 4, 2, 1) weight, [2, Cout, 4, 4] -> bf16 [chunks][3][2][16][32] with
    n = 2 (4 ky + kx) + o (csrc/engine_small.hip `flow_head_planes_fwd_mfma<1>`)."""
    cout = weight2.shape[1]
    chunks = ig.pad32(cout) // 32
    w = torch.zeros(chunks * 32, 32, dtype=torch.float32, device=weight2.device)          # [c][n]
    w[:cout] = weight2.detach().float().reshape(2, cout, 16).permute(1, 2, 0).reshape(cout, 32)
    wn = w.view(chunks, 32, 32).permute(0, 2, 1).contiguous()                               # [chunk][n][c]
    return ig._split3(wn).view(3, chunks, 32, 32).permute(1, 0, 2, 3).contiguous()


class FlowNetCHeadEngine:
    def __init__(self, net, B: int, H: int, W: int, device):
        if H % 64 or W % 64:
            raise ValueError("FlowNetC head engine: frame sides must be multiples of 64")
        L.lib()
        self.net, self.B, self.H, self.W, self.dev = net, int(B), int(H), int(W), torch.device(device)
        g = {s: (H // s, W // s) for s in (4, 8, 16, 32, 64)}
        self.grid = g
        P = lambda s, chunks: ig.Planes(B, g[s][0], g[s][1], chunks, self.dev)
        G = lambda s, chunks: ig.GradSum(B, g[s][0], g[s][1], chunks, self.dev)
        # FlowNetC (siamese: correlation + conv_redir in front of conv3_1) or the plain FlowNetS trunk of FlowNet2 / FlowNet2S
        # (models/flownet2/FlowNetS.py:15-104: conv3_1 reads conv3 directly; same layers and names behind it)
        self.siamese = hasattr(net, "conv_redir")
        # ---- activations
        self.c3a_p, self.c3b_p = (P(8, 8), P(8, 8)) if self.siamese else (None, None)
        self.in31, self.cat3, self.cat2 = P(8, 15 if self.siamese else 8), P(8, 13), P(4, 7)
        self.c4a, self.cat4 = P(16, 16), P(16, 25)
        self.c5a, self.cat5 = P(32, 16), P(32, 33)
        self.c6a, self.c6 = P(64, 32), P(64, 32)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.flow = {k: torch.zeros(B, 2, *g[2 ** k], **f32) for k in (6, 5, 4, 3, 2)}
        # ---- gradients
        self.G_cat2, self.G_cat3, self.G_cat4, self.G_cat5 = G(4, 7), G(8, 13), G(16, 25), G(32, 33)
        self.G_c6, self.G_in31, self.G_c3a = G(64, 32), G(8, 15 if self.siamese else 8), G(8, 8)
        self.gz_cat2, self.gz_cat3, self.gz_cat4, self.gz_cat5 = P(4, 7), P(8, 13), P(16, 25), P(32, 33)
        self.gz_c6, self.gz_c6a, self.gz_c5a, self.gz_c4a = P(64, 32), P(64, 32), P(32, 16), P(16, 16)
        self.gz_in31 = P(8, 15) if self.siamese else None
        self.g_flow = {k: torch.zeros(B, 2, *g[2 ** k], **f32) for k in (6, 5, 4, 3)}
        self.g_corr = torch.zeros(B, 21, 21, *g[8], **f32)
        self.g_c2a = torch.zeros(B, 128, *g[4], **f32)
        self.g_c3a, self.g_c3a_redir, self.g_c3b = (torch.zeros(B, 256, *g[8], **f32) for _ in range(3))
        self.c3_nchw = torch.zeros(2 * B, 256, *g[8], **f32)        # both frames' conv3 for the correlation kernels
        self._prefix = None                                          # full-frame conv1-3 buffers + launches, built on first use
        self.flow_out, self.flow_scale = self.flow[2], float(getattr(net, "div_flow", 20.0))   # what the step's fused loss kernel reads
        self._wprefixes, self._wprefix = {}, None                    # window-prefix state per window size; the one used last
        self._build_launches()

    # ------------------------------------------------------------------------------------------------ set-up
    def _conv(self, name):
        return getattr(self.net, name)[0]

    def _build_launches(self):
        net, g, B = self.net, self.grid, self.B
        plans = []      # (weights, make_launch kwargs without ws) -> sized together for one split-K workspace

        def plan(wi, x, in_chunk0, rows, out_hw, **kw):
            M = B * rows[0] * rows[1]
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            kw.setdefault("variant", self._variant_for(wi))
            bm, target = self._tile_rows_and_slots(wi, kw)
            S = ig.splitk_for(M, wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target)
            kw["variant"], S = ig.tuned(wi, M, kw, kw["variant"], S, rows=rows)
            plans.append((wi, x, in_chunk0, rows, out_hw, S, kw))
            return len(plans) - 1

        # 128-column launches: csrc/igemm.hip variant 6 (ping-pong: 256 x 128 tiles, two wave groups of one workgroup per CU half a
        # step apart); UFR_IGEMM_PIPE=5: pipelined 128 x 128, two workgroups per CU; UFR_IGEMM_PIPE=0: the single-stage kernel.
        # Same box, one iteration: 6.27 / 6.56 / 6.88 ms (profiles/r2_bench_pingpong_ab.txt).  64-column launches (deconv2
        # forward, conv1, conv_redir) always run the single-stage 128 x 64 tile.
        self._pipe_variant = {"0": 2, "5": 5}.get(os.environ.get("UFR_IGEMM_PIPE", "6"), 6)

        fwd, bwd = {}, {}
        cw = lambda n, s, p: ig.conv_forward_weights(self._conv(n).weight, s, p)
        cb = lambda n, s, p: ig.conv_backward_weights(self._conv(n).weight, s, p)
        bias = lambda n: self._conv(n).bias.detach().float().contiguous()
        # forward chain (FlowNetC.py:142-160)
        if self.siamese:
            fwd["conv_redir"] = plan(cw("conv_redir", 1, 0), self.c3a_p, 0, g[8], g[8], out_planes=self.in31, out_chunk0=0,
                                     bias=bias("conv_redir"))
        fwd["conv3_1"] = plan(cw("conv3_1", 1, 1), self.in31, 0, g[8], g[8], out_planes=self.cat3, out_chunk0=0, bias=bias("conv3_1"))
        fwd["conv4"] = plan(cw("conv4", 2, 1), self.cat3, 0, g[16], g[16], out_planes=self.c4a, bias=bias("conv4"))
        fwd["conv4_1"] = plan(cw("conv4_1", 1, 1), self.c4a, 0, g[16], g[16], out_planes=self.cat4, out_chunk0=0, bias=bias("conv4_1"))
        fwd["conv5"] = plan(cw("conv5", 2, 1), self.cat4, 0, g[32], g[32], out_planes=self.c5a, bias=bias("conv5"))
        fwd["conv5_1"] = plan(cw("conv5_1", 1, 1), self.c5a, 0, g[32], g[32], out_planes=self.cat5, out_chunk0=0, bias=bias("conv5_1"))
        fwd["conv6"] = plan(cw("conv6", 2, 1), self.cat5, 0, g[64], g[64], out_planes=self.c6a, bias=bias("conv6"))
        fwd["conv6_1"] = plan(cw("conv6_1", 1, 1), self.c6a, 0, g[64], g[64], out_planes=self.c6, bias=bias("conv6_1"))
        # refinement (FlowNetC.py:162-183): deconvK reads the level above, writes the middle segment of catK
        dec = {5: (self.c6, 64, self.cat5, 16), 4: (self.cat5, 32, self.cat4, 16), 3: (self.cat4, 16, self.cat3, 8),
               2: (self.cat3, 8, self.cat2, 4)}
        for k, (src, s_in, dst, chunk0) in dec.items():
            n = f"deconv{k}"
            fwd[n] = plan(ig.deconv_forward_weights(self._conv(n).weight, 1), src, 0, g[s_in], g[s_in // 2], out_planes=dst,
                          out_chunk0=chunk0, bias=bias(n))
        # backward (reverse order of use; see module docstring)
        gz = {5: (self.gz_cat5, 16, 16), 4: (self.gz_cat4, 16, 8), 3: (self.gz_cat3, 8, 4), 2: (self.gz_cat2, 4, 2)}
        Gabove = {2: self.G_cat3, 3: self.G_cat4, 4: self.G_cat5}
        # deconvK's input ends with the two channels of the upsampled flow: 386 / 770 / 1026 columns would pad the GEMM to
        # 512 / 896 / 1152 (and deconv3 / deconv4 to a second round of workgroups); the 384 / 768 / 1024 feature channels go
        # through the igemm and the two flow columns through the 2-channel kernel (`_deconv_tail`)
        self.tail_w, self.tail_args = {}, {}
        for k in (2, 3, 4):
            src, chunk0, _ = gz[k]
            s_in = {2: 8, 3: 16, 4: 32}[k]
            w = self._conv(f"deconv{k}").weight
            main = w.shape[0] - 2
            bwd[f"deconv{k}"] = plan(ig.deconv_backward_weights(w[:main], 1), src, chunk0, g[s_in], g[s_in], out_f32=Gabove[k])
            self.tail_w[k] = _pack_flow_tail_mfma(w[main:])
            self.tail_args[k] = (src, chunk0, ig.pad32(w.shape[1]) // 32, Gabove[k], main // 32, g[s_in])
        bwd["deconv5"] = plan(ig.deconv_backward_weights(self._conv("deconv5").weight, 1), self.gz_cat5, 16, g[64], g[64],
                              add=self.G_c6, mask=self.c6, out_planes=self.gz_c6)
        bwd["conv6_1"] = plan(cb("conv6_1", 1, 1), self.gz_c6, 0, g[64], g[64], mask=self.c6a, out_planes=self.gz_c6a)
        bwd["conv6"] = plan(cb("conv6", 2, 1), self.gz_c6a, 0, g[64], g[32], add=self.G_cat5, mask=self.cat5, out_planes=self.gz_cat5)
        bwd["conv5_1"] = plan(cb("conv5_1", 1, 1), self.gz_cat5, 0, g[32], g[32], mask=self.c5a, out_planes=self.gz_c5a)
        bwd["conv5"] = plan(cb("conv5", 2, 1), self.gz_c5a, 0, g[32], g[16], add=self.G_cat4, mask=self.cat4, out_planes=self.gz_cat4)
        bwd["conv4_1"] = plan(cb("conv4_1", 1, 1), self.gz_cat4, 0, g[16], g[16], mask=self.c4a, out_planes=self.gz_c4a)
        bwd["conv4"] = plan(cb("conv4", 2, 1), self.gz_c4a, 0, g[16], g[8], add=self.G_cat3, mask=self.cat3, out_planes=self.gz_cat3)
        if self.siamese:
            # its 473 columns: chunk 0 (conv_redir's output) is read as planes by conv_redir's adjoint, chunks 1 .. (the cost volume)
            # as fp32 by the correlation's: each leaves the tile once, in the form its reader wants (4.4 instead of 10 B per element)
            bwd["conv3_1"] = plan(cb("conv3_1", 1, 1), self.gz_cat3, 0, g[8], g[8], mask=self.in31, out_planes=self.gz_in31,
                                  out_f32=self.G_in31, planes_chunks=1, f32_first_chunk=1)
            bwd["conv_redir"] = plan(cb("conv_redir", 1, 0), self.gz_in31, 0, g[8], g[8], out_f32=self.G_c3a)
        else:                                   # the trunk's input IS conv3's activation: its gradient leaves unmasked
            bwd["conv3_1"] = plan(cb("conv3_1", 1, 1), self.gz_cat3, 0, g[8], g[8], out_f32=self.G_in31)
        need = max([len(wi.phases) * S * B * rows[0] * rows[1] * wi.Npad for wi, _, _, rows, _, S, _ in plans if S > 1] + [1])
        self.ws = torch.empty(need, dtype=torch.float32, device=self.dev)
        launches = [ig.make_launch(wi, x, c0, rows, out_hw, splitk=S, ws=self.ws if S > 1 else None, **kw)
                    for wi, x, c0, rows, out_hw, S, kw in plans]
        self.fwd = {k: launches[i] for k, i in fwd.items()}
        self.bwd = {k: launches[i] for k, i in bwd.items()}
        self._plans = {("fwd", k): plans[i] for k, i in fwd.items()}
        self._plans.update({("bwd", k): plans[i] for k, i in bwd.items()})
        self._band, self.fwd_band, self.bwd_band = None, {}, {}
        # 2-channel layers
        self.pf_w = {k: _pack_flow_head(getattr(net, f"predict_flow{k}").weight) for k, _ in _HEADS}
        self.pf_wm = {k: _pack_flow_head_mfma(getattr(net, f"predict_flow{k}").weight) for k, _ in _HEADS}
        self.pf_b = {k: getattr(net, f"predict_flow{k}").bias.detach().float().contiguous() for k, _ in _HEADS}
        self.up = {k: getattr(net, f"upsampled_flow{k}_to_{k - 1}") for k in (6, 5, 4, 3)}
        self.up_w = {k: m.weight.detach().float().contiguous() for k, m in self.up.items()}
        self.up_b = {k: (m.bias.detach().float().contiguous() if m.bias is not None else None) for k, m in self.up.items()}
        self.pf_src = {6: (self.c6, 32), 5: (self.cat5, 33), 4: (self.cat4, 25), 3: (self.cat3, 13), 2: (self.cat2, 7)}
        self.pf_G = {6: self.G_c6, 5: self.G_cat5, 4: self.G_cat4, 3: self.G_cat3, 2: self.G_cat2}
        # upsampled_flowK_to_K-1 writes the last chunk of cat(K-1)
        self.up_dst = {6: (self.cat5, 32), 5: (self.cat4, 24), 4: (self.cat3, 12), 3: (self.cat2, 6)}
        self.up_G = {6: (self.G_cat5, 32), 5: (self.G_cat4, 24), 4: (self.G_cat3, 12), 3: (self.G_cat2, 6)}

    # ------------------------------------------------------------------------------------------------ conv1-3
    def _variant_for(self, wi):
        """The igemm form of a launch: the 128-column forms as `UFR_IGEMM_PIPE` says, 64-column launches single-stage."""
        return self._pipe_variant if wi.Npad % 128 == 0 else 2

    @staticmethod
    def _tile_rows_and_slots(wi, kw):
        """(tile rows, resident workgroups on the chip) of the kernel a launch runs on: what split-K is sized against."""
        v = kw.get("variant", 0)
        if wi.Npad % 128:
            return 128, 768                    # single-stage 128 x 64 tiles, three workgroups per CU
        if v == 4:
            return 64, 1024                    # 64 x 128 tiles, four workgroups per CU
        if v == 5:
            return 128, 512                    # register-held fragments: two workgroups per CU
        if v == 6:
            return 256, 256                    # ping-pong: one 256-row workgroup per CU
        return 128, 768

    def _build_prefix(self):
        """Full-frame conv1-3 for both frames of every pair (models/FlowNetC.py:100-119), run once per attack() call, as two
        chains of three igemm launches -- one per frame set -- that write where the head reads: the first frames' conv2 into
        cat2's chunks 0-3 (the skip connection) and their conv3 into c3a_p, the second frames' conv3 into c3b_p.  (One chain
        over both sets needed three strided copies afterwards, 0.2 ms per call; with 256-row tiles the halves fill the same
        number of rounds: conv2 4 + 4 for 8, conv3 2 + 2 for 4.)"""
        B, dev = self.B, self.dev
        H, W = self.H, self.W
        bias = lambda n: self._conv(n).bias.detach().float().contiguous()
        w2 = ig.conv_forward_weights(self._conv("conv2").weight, 2, 2)
        w3 = ig.conv_forward_weights(self._conv("conv3").weight, 2, 2)
        # conv2 (K = 25 taps x 2 chunks) at 2 x 8 frames: 1.35 ms on single-stage 128 x 128 tiles, 1.02-1.05 on 64 x 128 tiles
        # (four workgroups per CU), 0.99-1.00 on the pipelined 128 x 128 kernel, 0.89-0.94 on the ping-pong kernel
        v2 = v3 = self._pipe_variant
        halves = {}
        for h, c2_dst, c3_dst in (("a", self.cat2, self.c3a_p), ("b", ig.Planes(B, H // 4, W // 4, 4, dev), self.c3b_p)):
            c1 = ig.Planes(B, H // 2, W // 2, 2, dev)
            l2 = ig.make_launch(w2, c1, 0, (H // 4, W // 4), (H // 4, W // 4), out_planes=c2_dst, out_chunk0=0, bias=bias("conv2"),
                                variant=v2)
            l3 = ig.make_launch(w3, c2_dst, 0, (H // 8, W // 8), (H // 8, W // 8), out_planes=c3_dst, out_chunk0=0,
                                bias=bias("conv3"), variant=v3)
            P = dict(c1=c1, conv2=l2, conv3=l3, conv2_wi=w2, conv3_wi=w3, b1=bias("conv1"), w1=self._conv("conv1").weight.detach())
            P.update(self._conv1_launch(B, H, W, c1))
            halves[h] = P
        self._prefix = halves

    def _conv1_launch(self, n: int, H: int, W: int, c1: ig.Planes) -> dict:
        """conv1 = Conv2d(3, 64, 7, 2, 3) + bias + LeakyReLU as an igemm launch over the packed planes of the raw frames
        (csrc/plane_layout.hip `conv1_pack_kernel`: pixel-unshuffle + two columns per chunk -> 8 taps of one chunk, the mean
        subtraction and the zero padding inside the buffer), writing conv1's planes directly."""
        bias = self._conv("conv1").bias.detach().float().contiguous()
        if os.environ.get("UFR_CONV1_DIRECT", "1") != "0":
            # round 4: ONE kernel from the raw frames to conv1's planes (csrc/conv1_direct.hip): no packed buffer, no pack pass
            conv1 = self._conv("conv1")
            return dict(direct=dict(wimg=ig.conv1_direct_weights(conv1.weight), bias=bias, c1=c1, n=n, hw=(H, W),
                                    gflop=2.0 * n * (H // 2) * (W // 2) * 147 * 64 / 1e9))
        packed = ig.Planes(n, H // 2 + 3, W // 2 + 2, 1, self.dev)
        wi = ig.conv1_packed_weights(self._conv("conv1").weight)
        launch = ig.make_launch(wi, packed, 0, (H // 2, W // 2), (H // 2, W // 2), out_planes=c1, bias=bias, variant=2)
        return dict(packed=packed, conv1=launch, conv1_wi=wi)

    def _conv1(self, P: dict, a: torch.Tensor, b: torch.Tensor | None):
        """conv1 + bias + LeakyReLU of one or two raw frame stacks into P['c1']."""
        a = a.contiguous()
        L.require_hip(a, "frames")
        nb = 0 if b is None else int(b.shape[0])
        mean = self.net._mean64.reshape(-1).contiguous()
        D = P.get("direct")
        if D is not None:
            c1 = D["c1"]
            L.check(L.lib().ufr_conv1_direct(L.ptr(a), L.ptr(b.contiguous()) if b is not None else None, int(a.shape[0]), nb,
                                             int(a.shape[2]), int(a.shape[3]), L.ptr(mean), L.ptr(D["wimg"]), L.ptr(D["bias"]),
                                             float(ig.LEAKY), L.ptr(c1.t), c1.plane_stride, 0, L.stream()), "conv1 direct")
            return
        pk = P["packed"]
        L.check(L.lib().ufr_conv1_pack_planes(L.ptr(a), L.ptr(b.contiguous()) if b is not None else None, L.ptr(pk.t),
                                              pk.plane_stride, int(a.shape[0]), nb, int(a.shape[2]), int(a.shape[3]),
                                              L.ptr(mean), L.stream()), "conv1 pack")
        P["conv1"]()

    def prefix_full(self, frames_a: torch.Tensor, frames_b: torch.Tensor):
        """New frames: conv2 of the first frames and conv3 of both, for the full frame, straight into the head's plane
        buffers (cat2[0:4], c3a_p, c3b_p) plus conv3 in NCHW for the correlation kernels."""
        if self._prefix is None:
            self._build_prefix()
        B = self.B

        def chain(h, frames, out):
            P = self._prefix[h]
            self._conv1(P, frames, None)
            P["conv2"]()
            P["conv3"]()
            (self.c3a_p if h == "a" else self.c3b_p).to_nchw(256, 0, out=out)
        # (the two chains on two streams, so that one launch's tail is filled by the other chain's workgroups, measured no
        # faster: 5.638 against 5.620 ms in one call, gpurun r4_call5 -- the chains stay on one stream)
        chain("a", frames_a, self.c3_nchw[:B])
        chain("b", frames_b, self.c3_nchw[B:])
        self._c3a, self._c3b = self.c3_nchw[:B], self.c3_nchw[B:]

    def load_prefix_features(self, c2_all: torch.Tensor, c3_all: torch.Tensor):
        """The same from NCHW features the caller already holds (train()'s clean forward seeds the attack's cache)."""
        B = self.B
        self.cat2.load_nchw(c2_all[:B].contiguous(), 0)
        self.c3_nchw.copy_(c3_all[:2 * B])
        self._c3a, self._c3b = self.c3_nchw[:B], self.c3_nchw[B:]
        self.c3a_p.load_nchw(self._c3a, 0)
        self.c3b_p.load_nchw(self._c3b, 0)

    def scatter_window_features(self, c2_w: torch.Tensor, c3_w: torch.Tensor, win: torch.Tensor, wh: int, ww: int, m2: int, m3: int):
        """The windowed prefix's conv2 (first frames) / conv3 (both frames) patched into the cached planes and into the NCHW
        conv3 the correlation reads; `win` = the step's origin table, m2 / m3 = rim margins in cells (cone.py)."""
        lib, B, st = L.lib(), self.B, L.stream()
        h4, w4 = self.grid[4]
        h8, w8 = self.grid[8]
        L.check(lib.ufr_window_scatter_planes(L.ptr(c2_w), L.ptr(self.cat2.t), self.cat2.plane_stride, 0, L.ptr(win), B, B, 128,
                                              h4, w4, wh // 4, ww // 4, 4, m2, st), "window -> planes (conv2)")
        for k, dst in ((0, self.c3a_p), (1, self.c3b_p)):
            L.check(lib.ufr_window_scatter_planes(L.ptr(c3_w[k * B:]), L.ptr(dst.t), dst.plane_stride, 0, L.ptr(win), B, B, 256,
                                                  h8, w8, wh // 8, ww // 8, 8, m3, st), "window -> planes (conv3)")
        L.check(lib.ufr_window_scatter(L.ptr(c3_w), L.ptr(self.c3_nchw), L.ptr(win), B, 2 * B, 256, h8, w8, wh // 8, ww // 8, 8,
                                       m3, st), "window -> nchw (conv3)")

    # ------------------------------------------------------------------------------------------------ windowed conv1-3
    def _build_window_prefix(self, wh: int, ww: int):
        """conv1-3 of both frames on the patch attack's prefix window (patch_attack.py `_forward_cone`: [2B, 3, wh, ww]) and
        their data gradients, all on the igemm: conv2 / conv3 (5x5, stride 2: 92% of the window's FLOPs) as they are, conv1
        (3 input channels) over the packed planes of the raw window stack (`_conv1_launch`), its data gradient with respect
        to those planes followed by the unpacking.  One state per window size, kept for the engine's life: a step's captured
        HIP graphs hold raw pointers into these buffers, and another step on the same network may use another size."""
        B2, dev = 2 * self.B, self.dev
        f32 = dict(dtype=torch.float32, device=dev)
        h2, w2, h4, w4, h8, w8 = wh // 2, ww // 2, wh // 4, ww // 4, wh // 8, ww // 8
        c1, c2, c3 = ig.Planes(B2, h2, w2, 2, dev), ig.Planes(B2, h4, w4, 4, dev), ig.Planes(B2, h8, w8, 8, dev)
        gz_c3, gz_c2 = ig.Planes(B2, h8, w8, 8, dev), ig.Planes(B2, h4, w4, 4, dev)
        gz_c1, G_p = ig.Planes(B2, h2, w2, 2, dev), ig.GradSum(B2, h2 + 3, w2 + 2, 1, dev)
        G_gw2 = ig.GradSum(B2, h4, w4, 4, dev)       # the conv2 tap's window gradient (first frames; second frames stay 0)
        bias = lambda n: self._conv(n).bias.detach().float().contiguous()
        plans = [
            # (the window's forward convolutions: 64 x 128 tiles measured 1.3x faster than 128 x 128 at the same split,
            # 0.047 vs 0.055 ms)
            (ig.conv_forward_weights(self._conv("conv2").weight, 2, 2), c1, (h4, w4), (h4, w4),
             dict(out_planes=c2, bias=bias("conv2"), variant=4)),
            (ig.conv_forward_weights(self._conv("conv3").weight, 2, 2), c2, (h8, w8), (h8, w8),
             dict(out_planes=c3, bias=bias("conv3"), variant=4)),
            # conv3's data gradient + the skip connection's gradient, x LeakyReLU'(conv2) -> conv2's gradient planes
            (ig.conv_backward_weights(self._conv("conv3").weight, 2, 2), gz_c3, (h8, w8), (h4, w4),
             dict(add=G_gw2, mask=c2, out_planes=gz_c2)),
            (ig.conv_backward_weights(self._conv("conv2").weight, 2, 2), gz_c2, (h4, w4), (h2, w2),
             dict(out_planes=gz_c1, mask=c1)),
            # conv1's data gradient with respect to the packed planes (the unpacking follows)
            (ig.conv1_packed_backward_weights(self._conv("conv1").weight), gz_c1, (h2 + 3, w2 + 2), (h2 + 3, w2 + 2),
             dict(out_f32=G_p)),
        ]
        sized = []
        for wi, x, rows, out_hw, kw in plans:
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            kw.setdefault("variant", self._variant_for(wi))
            bm, target = self._tile_rows_and_slots(wi, kw) if kw.get("variant") in (5, 6) else (128, 768)
            sized.append(ig.splitk_for(B2 * rows[0] * rows[1], wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target))
        need = max([len(wi.phases) * S * B2 * rows[0] * rows[1] * wi.Npad for (wi, _, rows, _, _), S in zip(plans, sized) if S > 1] + [1])
        ws = torch.empty(need, **f32)
        launches = [ig.make_launch(wi, x, 0, rows, out_hw, splitk=S, ws=ws if S > 1 else None, **kw)
                    for (wi, x, rows, out_hw, kw), S in zip(plans, sized)]
        wis = {k + "_wi": p[0] for k, p in zip(("conv2", "conv3", "conv3_bwd", "conv2_bwd", "conv1_bwd"), plans)}
        P = dict(**wis, hw=(wh, ww), c1=c1, c2=c2, c3=c3, gz_c3=gz_c3, gz_c2=gz_c2, G_gw2=G_gw2, ws=ws,
                 conv2=launches[0], conv3=launches[1], conv3_bwd=launches[2], conv2_bwd=launches[3], conv1_bwd=launches[4],
                 G_p=G_p, gxw=torch.zeros(B2, 3, wh, ww, **f32),
                 c2_nchw=torch.zeros(B2, 128, h4, w4, **f32), c3_nchw=torch.zeros(B2, 256, h8, w8, **f32))
        P.update(self._conv1_launch(B2, wh, ww, c1))
        self._wprefixes[(wh, ww)] = P
        self._bound_wprefixes()
        return P

    MAX_WINDOW_PREFIXES = 4

    def _bound_wprefixes(self):
        """At most MAX_WINDOW_PREFIXES window sizes stay cached (least recently used goes).  A step that captured graphs over a
        state holds the state itself (PatchAttackStep._wp_hold), so dropping it here never frees memory a graph points into."""
        while len(self._wprefixes) > self.MAX_WINDOW_PREFIXES:
            del self._wprefixes[next(iter(self._wprefixes))]

    def window_prefix(self, wh: int, ww: int) -> dict:
        """The window-prefix state for a (wh, ww) window; `_wprefix` = the one used last (launch_table, the backward)."""
        P = self._wprefixes.pop((int(wh), int(ww)), None)
        if P is None:
            P = self._build_window_prefix(int(wh), int(ww))
        else:
            self._wprefixes[(int(wh), int(ww))] = P          # most recently used last
        self._wprefix = P
        return P

    def window_prefix_forward(self, xw: torch.Tensor, win: torch.Tensor, m2: int, m3: int):
        """conv1-3 of the window stack `xw` [2B, 3, wh, ww] (raw frames; first frames, then second frames), patched into the
        cached full-frame features.  The window's convolutions zero-pad at the window's edges exactly like the torch
        prefix they replace; the inexact rim (m2 / m3 cells) is skipped by the scatter."""
        wh, ww = int(xw.shape[2]), int(xw.shape[3])
        P = self.window_prefix(wh, ww)
        self._conv1(P, xw.detach(), None)
        P["conv2"]()
        P["conv3"]()
        P["c2"].to_nchw(128, 0, out=P["c2_nchw"])                         # (the scatter kernels take NCHW windows)
        P["c3"].to_nchw(256, 0, out=P["c3_nchw"])
        self.scatter_window_features(P["c2_nchw"], P["c3_nchw"], win, wh, ww, m2, m3)

    def window_gather_conv2_gradient(self, win: torch.Tensor, m2: int):
        """The conv2 tap's window gradient straight from the head's chunk-major sum of cat2 (chunks 0-3 = conv2 of the
        first frames) into the addend of conv3's data gradient: no full-frame NCHW conversion, no NCHW window."""
        P = self._wprefix
        wh, ww = P["hw"]
        h4, w4 = self.grid[4]
        L.check(L.lib().ufr_window_gather_chunks(L.ptr(self.G_cat2.t), L.ptr(P["G_gw2"].t), L.ptr(win), self.B, self.B, 2 * self.B, 4,
                                                 h4, w4, wh // 4, ww // 4, 4, int(m2), L.stream()), "window gather (chunks)")

    def window_prefix_backward(self, gw3: torch.Tensor, gw2: torch.Tensor | None = None) -> torch.Tensor:
        """d loss / d xw from the window gradients of the two taps: gw3 [2B, 256, wh/8, ww/8] (conv3 of both frames, rim
        zeroed); the conv2 tap's (first frames) was put into `G_gw2` by `window_gather_conv2_gradient`, or is handed over
        here as NCHW [B, 128, wh/4, ww/4]."""
        P, B = self._wprefix, self.B
        wh, ww = P["hw"]
        if gw2 is not None:                      # NCHW -> chunk-major addend (tests, callers without the engine's sums)
            full = torch.zeros(2 * B, 128, wh // 4, ww // 4, dtype=torch.float32, device=self.dev)
            full[:B] = gw2
            P["G_gw2"].t.copy_(full.view(2 * B, 4, 32, wh // 4, ww // 4).permute(1, 0, 3, 4, 2).reshape(4, -1, 32))
        gz = P["gz_c3"]                          # x LeakyReLU'(conv3), split into the planes conv3's data gradient reads
        L.check(L.lib().ufr_nchw_grad_to_planes(L.ptr(gw3), L.ptr(P["c3_nchw"]), L.ptr(gz.t), gz.plane_stride, 0, 2 * B, 256, wh // 8,
                                                ww // 8, ig.LEAKY, L.stream()), "window gradient -> planes")
        P["conv3_bwd"]()                         # + G_gw2, x LeakyReLU'(conv2) -> gz_c2 (epilogue)
        P["conv2_bwd"]()                         # x LeakyReLU'(conv1) -> conv1's gradient planes
        P["conv1_bwd"]()                         # -> gradient of the packed planes -> gradient of the raw window stack
        L.check(L.lib().ufr_conv1_unpack_grad(L.ptr(P["G_p"].t), L.ptr(P["gxw"]), 2 * B, wh, ww, L.stream()), "conv1 unpack")
        return P["gxw"]

    # ------------------------------------------------------------------------------------------------ column band
    # (level stride of the ROW grid, of the input grid) of the launches that run on the band's columns only
    _FWD_BAND = {"conv_redir": (8, 8), "conv3_1": (8, 8), "conv4": (16, 8), "conv4_1": (16, 16)}
    # data gradients on the band (band_conv.py's BAND_LAYERS): rows = the gy grid; the input (a gradient that itself exists
    # on the band only) reads as zero outside it
    _BWD_BAND = {"conv5": (32, None), "conv4_1": (16, 16), "conv4": (16, 16), "conv3_1": (8, 8), "conv_redir": (8, 8)}

    def attach_band(self, band):
        """Derive the banded launches for `band` (band_conv.Band: device-resident first column per pair, static width):
        the same descriptors with the tile rows restricted to the band -- taps read the full-frame planes, so there is no
        gather / scatter copy and every computed column is exact."""
        if self._band is band:
            return
        self._band, self.fwd_band, self.bwd_band = band, {}, {}
        if not band.width:
            return
        W = self.W
        if band.width % 32 or band.width > W:
            raise ValueError("band width must be a multiple of 32 pixels inside the frame")
        origin = band.win[:, 1]                                   # int32 view, 8 elements per pair

        def derive(kind, name, ls_rows, ls_in):
            wi, x, c0, rows, out_hw, S, kw = self._plans[(kind, name)]
            rows_b = (rows[0], band.width // ls_rows)
            extra = dict(row_band=(origin, 8, ls_rows))
            if ls_in is not None:
                extra["in_band"] = (origin, 8, ls_in, band.width // ls_in)
            M = self.B * rows_b[0] * rows_b[1]
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            bm, target = self._tile_rows_and_slots(wi, kw)
            Sb = ig.splitk_for(M, wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target)
            if len(wi.phases) * Sb * M * wi.Npad > self.ws.numel():
                Sb = 1
            return ig.make_launch(wi, x, c0, rows_b, out_hw, splitk=Sb, ws=self.ws if Sb > 1 else None, **kw, **extra)

        for name, (ls_rows, _) in self._FWD_BAND.items():
            if ("fwd", name) in self._plans:
                self.fwd_band[name] = derive("fwd", name, ls_rows, None)   # forward inputs are valid everywhere
        for name, (ls_rows, ls_in) in self._BWD_BAND.items():
            if ("bwd", name) in self._plans:
                self.bwd_band[name] = derive("bwd", name, ls_rows, ls_in)

    def replan(self, kind: str, name: str, tag: str):
        """(weights, input planes, first chunk, row grid, output grid, make_launch kwargs) of a prepared launch, band geometry
        included: tools/sweep_igemm_launches.py rebuilds it with other kernel forms / split-K factors."""
        wi, x, c0, rows, out_hw, _, kw = self._plans[(kind, name)]
        kw = dict(kw)
        if tag == "band":
            band = self._band
            ls_rows, ls_in = (self._FWD_BAND[name][0], None) if kind == "fwd" else self._BWD_BAND[name]
            origin = band.win[:, 1]
            rows = (rows[0], band.width // ls_rows)
            kw["row_band"] = (origin, 8, ls_rows)
            if ls_in is not None:
                kw["in_band"] = (origin, 8, ls_in, band.width // ls_in)
        return wi, x, c0, rows, out_hw, kw

    def launch_table(self):
        """Every prepared igemm launch with its algorithmic work, for bench.py's per-kernel rooflines:
        [(name, 'fwd' | 'bwd', 'full' | 'band', launch, GFLOP)]."""
        rows = []
        for kind, table, tag in (("fwd", self.fwd, "full"), ("bwd", self.bwd, "full"), ("fwd", self.fwd_band, "band"),
                                 ("bwd", self.bwd_band, "band")):
            for name, launch in table.items():
                wi = self._plans[(kind, name)][0]
                d = launch.desc
                rows.append((name, kind, tag, launch, wi.flops(d.B * d.Hr * d.Wr) / 1e9))
        if self._prefix is not None:           # full-frame conv1-3 of the first / second frames, once per attack() call
            for h, suffix in (("a", ""), ("b", " 2nd frames")):
                F = self._prefix[h]
                for key in ("conv1", "conv2", "conv3"):
                    if key not in F:
                        continue
                    d = F[key].desc
                    rows.append((key + suffix, "fwd", "prefix", F[key], F[key + "_wi"].flops(d.B * d.Hr * d.Wr) / 1e9))
        P = getattr(self, "_wprefix", None)
        if P is not None:                      # conv2 / conv3 of the attack's prefix window (every iteration)
            for name, kind, key in (("conv1", "fwd", "conv1"), ("conv2", "fwd", "conv2"), ("conv3", "fwd", "conv3"),
                                    ("conv3", "bwd", "conv3_bwd"), ("conv2", "bwd", "conv2_bwd"), ("conv1", "bwd", "conv1_bwd")):
                if key not in P:
                    continue
                d = P[key].desc
                rows.append((name, kind, "window", P[key], P[key + "_wi"].flops(d.B * d.Hr * d.Wr) / 1e9))
        return rows

    def conv1_direct_table(self):
        """The direct conv1 launches (csrc/conv1_direct.hip) for bench.py's per-kernel rooflines: [(label, tag, fn, algorithmic
        bytes, GFLOP)] -- HBM-bound: the raw frames in, conv1's three bf16 planes out."""
        rows = []
        sets = []
        if self._prefix is not None:
            sets += [("conv1 direct" + sfx, "prefix", self._prefix[h]) for h, sfx in (("a", ""), ("b", " 2nd frames"))]
        if getattr(self, "_wprefix", None) is not None:
            sets.append(("conv1 direct", "window", self._wprefix))
        for label, tag, P in sets:
            D = P.get("direct")
            if D is None:
                continue
            n, (H, W) = D["n"], D["hw"]
            frames = torch.rand(n, 3, H, W, device=self.dev)
            nbytes = n * 3 * H * W * 4 + n * (H // 2) * (W // 2) * 64 * 6
            rows.append((label, tag, (lambda P=P, frames=frames: self._conv1(P, frames, None)), float(nbytes), D["gflop"]))
        return rows

    # ------------------------------------------------------------------------------------------------ small launches
    def _pf_forward(self, k):
        src, chunks = self.pf_src[k]
        L.check(L.lib().ufr_flow_head_planes_forward_mfma(L.ptr(src.t), src.plane_stride, 0, chunks, L.ptr(self.pf_wm[k]),
                                                          self.pf_wm[k].shape[0], L.ptr(self.pf_b[k]), L.ptr(self.flow[k]), self.B, src.H, src.W,
                                                          L.stream()), "predict_flow forward (mfma)")

    def _pf_backward(self, k, gy, accumulate, finalize=None):
        """predict_flowK^T; `finalize` = (activation planes, gradient planes, first chunk, chunks) of the segment whose gradient
        sum is complete with this launch (deconvK's output): x LeakyReLU' -> planes in the same kernel (round 4)."""
        src, chunks = self.pf_src[k]
        if finalize is not None:
            act, out, c0, n = finalize
            L.check(L.lib().ufr_flow_head_planes_backward_finalize(L.ptr(gy), L.ptr(self.pf_w[k]), self.pf_w[k].shape[0], L.ptr(self.pf_G[k].t),
                                                                   self.pf_G[k].chunks, 0, chunks, self.B,
                                                                   src.H, src.W, int(accumulate), L.ptr(act.t), L.ptr(out.t), out.plane_stride,
                                                                   int(c0), int(n), float(ig.LEAKY), L.stream()),
                    "predict_flow backward + finalize")
            return
        L.check(L.lib().ufr_flow_head_planes_backward(L.ptr(gy), L.ptr(self.pf_w[k]), self.pf_w[k].shape[0], L.ptr(self.pf_G[k].t),
                                                      self.pf_G[k].chunks, 0, chunks, self.B,
                                                      src.H, src.W, int(accumulate), L.stream()), "predict_flow backward")

    def _up_forward(self, k):
        dst, chunk = self.up_dst[k]
        f = self.flow[k]
        L.check(L.lib().ufr_flow_up_planes_forward(L.ptr(f), L.ptr(self.up_w[k]), L.ptr(self.up_b[k]) if self.up_b[k] is not None else None,
                                                   L.ptr(dst.t), dst.plane_stride, chunk, self.B, f.shape[2], f.shape[3],
                                                   L.stream()), "upsampled_flow forward")

    def _deconv_tail(self, k):
        src, chunk0, chunks, Gd, out_chunk, (h, w) = self.tail_args[k]
        L.check(L.lib().ufr_deconv_flow_tail_backward_mfma(L.ptr(src.t), src.plane_stride, chunk0, chunks, L.ptr(self.tail_w[k]),
                                                           self.tail_w[k].shape[0], L.ptr(Gd.t), Gd.chunks, out_chunk, self.B, h, w, L.stream()),
                "deconv data gradient, flow channels")

    def _up_backward(self, k):
        G, chunk = self.up_G[k]
        f = self.g_flow[k]
        L.check(L.lib().ufr_flow_up_planes_backward(L.ptr(G.t), chunk, L.ptr(self.up_w[k]), L.ptr(f), self.B, f.shape[2],
                                                    f.shape[3], 0, L.stream()), "upsampled_flow backward")

    def _finalize(self, Gs, mask, out, chunk0, chunks):
        L.check(L.lib().ufr_grad_finalize(L.ptr(Gs.t), chunk0, L.ptr(mask.t), chunk0, L.ptr(out.t), out.plane_stride, chunk0,
                                          Gs.M, chunks, ig.LEAKY, L.stream()), "gradient finalize")

    # ------------------------------------------------------------------------------------------------ the schedule
    def _forward_trunk(self, c2: torch.Tensor, c3: torch.Tensor) -> torch.Tensor:
        """FlowNetS: (conv2 [B,128,H/4,W/4], conv3 [B,256,H/8,W/8]) -> flow2; conv3_1 reads conv3 directly."""
        L.require_hip(c2, "c2")
        L.require_hip(c3, "c3")
        self.cat2.load_nchw(c2, 0)
        self.in31.load_nchw(c3, 0)
        for name in ("conv3_1", "conv4", "conv4_1", "conv5", "conv5_1", "conv6", "conv6_1"):
            self.fwd[name]()
        self._pf_forward(6)
        for k in (5, 4, 3, 2):
            self._up_forward(k + 1)
            self.fwd[f"deconv{k}"]()
            self._pf_forward(k)
        return self.flow[2]

    def forward_cached(self, band=None) -> torch.Tensor:
        """`forward` on the features the engine already holds (prefix_full / load_prefix_features + scatter_window_features)."""
        return self.forward(None, self._c3a, self._c3b, band)

    def forward(self, c2a: torch.Tensor | None, c3a: torch.Tensor, c3b: torch.Tensor, band=None) -> torch.Tensor:
        """(conv2 of frame 1 [B,128,H/4,W/4], conv3 of both frames [B,256,H/8,W/8]) -> flow2 [B,2,H/4,W/4].
        `band` (band_conv.Band) with `incremental` set: the features differ from the previous call's only inside the
        prefix window, so conv_redir / conv3_1 / conv4 / conv4_1 recompute the band's columns only -- the plane buffers
        still hold the previous iteration's activations everywhere else."""
        if not self.siamese:
            return self._forward_trunk(c2a, c3a)
        for t, name in ((c3a, "c3a"), (c3b, "c3b")):
            L.require_hip(t, name)
        self._c3a, self._c3b = c3a, c3b
        inc = False
        if band is not None:
            self.attach_band(band)
            inc = bool(band.width and band.incremental and band.inc_layers)
        if c2a is not None:                      # features handed over in NCHW: convert; None: the planes are up to date
            L.require_hip(c2a, "c2a")
            self.cat2.load_nchw(c2a, 0)
            self.c3a_p.load_nchw(c3a, 0)
            self.c3b_p.load_nchw(c3b, 0)
        # submodules.py:124-138 (`correlate`: /C) + FlowNetC.py:139 LeakyReLU fused into the correlation's epilogue:
        # matrix cores, planes in, conv3_1's input planes out (correlation_planes.hip)
        if inc and band.cone_win is not None and band.cone_hw[1] // 8 <= 23:
            # later iterations of a call: only the window's neighbourhood of the volume changes
            L.check(L.lib().ufr_corr_forward_planes_window(L.ptr(self.c3a_p.t), L.ptr(self.c3b_p.t), self.c3a_p.plane_stride,
                                                           L.ptr(self.in31.t), self.in31.plane_stride, 1, self.B, 256, *self.grid[8],
                                                           21, 2, 1.0 / 256.0, ig.LEAKY, L.ptr(band.cone_win), 8,
                                                           band.cone_hw[1] // 8, L.stream()), "correlation forward (planes, window)")
        else:
            L.check(L.lib().ufr_corr_forward_planes(L.ptr(self.c3a_p.t), L.ptr(self.c3b_p.t), self.c3a_p.plane_stride,
                                                    L.ptr(self.in31.t), self.in31.plane_stride, 1, self.B, 256, *self.grid[8], 21, 2,
                                                    1.0 / 256.0, ig.LEAKY, L.stream()), "correlation forward (planes)")
        for name in ("conv_redir", "conv3_1", "conv4", "conv4_1"):
            (self.fwd_band if inc else self.fwd)[name]()
        for name in ("conv5", "conv5_1", "conv6", "conv6_1"):
            self.fwd[name]()
        # refinement (FlowNetC.py:162-183).  (deconvK beside predict_flow(K+1) -> upsampled_flow(K+1) on a second stream inside the
        # graph -- they read cat(K+1) and write disjoint chunks of catK -- measured no faster: 5.773 against 5.729 ms in one call,
        # gpurun r4_call23; the same for the small launches of the backward beside deconvK's data gradient, 5.904 against 5.874,
        # r4_call13: the ping-pong igemm holds every CU's LDS, the small launches wait for its tail either way.  One stream.)
        self._pf_forward(6)
        for k in (5, 4, 3, 2):
            self._up_forward(k + 1)
            self.fwd[f"deconv{k}"]()
            self._pf_forward(k)
        return self.flow[2]

    def backward(self, g_flow2: torch.Tensor, band=None, fused_window: bool = True):
        """d loss / d flow2 -> (d/d conv2a, d/d conv3a, d/d conv3b), all NCHW float32 (static buffers).
        With a band: the data gradients of conv5, conv4_1, conv4, conv3_1 and conv_redir run on the band's columns and the
        correlation's adjoint on the window's cells (only those are read behind a windowed prefix).  When the band carries
        `g3_window` (the window-sized gradient of conv3, both frames: patch_attack.py) and `fused_window`, the correlation's
        adjoints are written there directly and the last two results are None."""
        L.require_hip(g_flow2, "g_flow2")
        B = self.B
        banded = band is not None and bool(band.width)
        if band is not None:
            self.attach_band(band)
        # level 2: predict_flow2 -> gradient of cat2 = [conv2a | deconv2 | flow3_up]
        gz = {2: (self.G_cat2, self.cat2, self.gz_cat2, 4, 2), 3: (self.G_cat3, self.cat3, self.gz_cat3, 8, 4),
              4: (self.G_cat4, self.cat4, self.gz_cat4, 16, 8), 5: (self.G_cat5, self.cat5, self.gz_cat5, 16, 16)}
        # predict_flowK's adjoint is the LAST contributor to deconvK's output gradient (deconv(K-1)'s data gradient and its two
        # flow columns come before it): LeakyReLU' and the split into gradient planes ride in its kernel instead of a
        # `grad_finalize` launch per level (UFR_FUSE_FINALIZE=0: the separate launches)
        fuse = os.environ.get("UFR_FUSE_FINALIZE", "1") != "0"
        fin = lambda k: (gz[k][1], gz[k][2], gz[k][3], gz[k][4]) if fuse else None
        self._pf_backward(2, g_flow2, accumulate=False, finalize=fin(2))
        eng_window = bool(fused_window and band is not None and getattr(band, "eng_window", False)
                          and getattr(self, "_wprefix", None) is not None)
        if eng_window:                           # the window prefix runs on the engine: its conv2 gradient stays chunk-major
            self.window_gather_conv2_gradient(band.cone_win, band.g2_margin)
        else:
            self.G_cat2.to_nchw(128, 0, out=self.g_c2a)
        for k in (2, 3, 4):
            Gs, act, out, chunk0, chunks = gz[k]
            if not fuse:
                self._finalize(Gs, act, out, chunk0, chunks)      # LeakyReLU' of deconvK's output
            self._up_backward(k + 1)                               # -> d/d flow(K+1)
            self.bwd[f"deconv{k}"]()                               # writes the gradient sum of cat(K+1)
            self._deconv_tail(k)
            self._pf_backward(k + 1, self.g_flow[k + 1], accumulate=True, finalize=fin(k + 1))
        Gs, act, out, chunk0, chunks = gz[5]
        if not fuse:
            self._finalize(Gs, act, out, chunk0, chunks)
        self._up_backward(6)
        self._pf_backward(6, self.g_flow[6], accumulate=False)      # G_c6 = predict_flow6^T; deconv5's adjoint adds to it
        for name in ("deconv5", "conv6_1", "conv6", "conv5_1"):
            self.bwd[name]()
        if not self.siamese:                    # FlowNetS trunk: conv3_1's data gradient IS d/d conv3
            for name in ("conv5", "conv4_1", "conv4", "conv3_1"):
                self.bwd[name]()
            self.G_in31.to_nchw(256, 0, out=self.g_c3a)
            return self.g_c2a, self.g_c3a, None
        for name in ("conv5", "conv4_1", "conv4", "conv3_1", "conv_redir"):
            (self.bwd_band if banded else self.bwd)[name]()
        # conv_redir's input and the correlation's two inputs
        gw = getattr(band, "g3_window", None) if (band is not None and fused_window) else None
        h8, w8 = self.grid[8]
        if gw is not None and band.cone_win is not None and band.cone_hw[1] // 8 <= 16 and w8 % 4 == 0:
            # windowed prefix behind the head: the cost volume's adjoints on the window's cells, on the matrix cores, straight
            # from the gradient sums into the window-sized gradient (correlation_window_mfma.hip)
            L.check(L.lib().ufr_corr_backward_window_fused(L.ptr(self._c3a), L.ptr(self._c3b), L.ptr(self.G_in31.t), 1, 1.0 / 256.0,
                                                           L.ptr(self.G_c3a.t), L.ptr(gw), B, 256, h8, w8, 21, 2,
                                                           L.ptr(band.cone_win), 8, band.cone_hw[0] // 8, band.cone_hw[1] // 8,
                                                           int(band.g3_margin), L.stream()), "correlation backward (window, fused)")
            return (None if eng_window else self.g_c2a), None, None
        self.G_c3a.to_nchw(256, 0, out=self.g_c3a_redir)
        self.G_in31.to_nchw(441, 1, scale=1.0 / 256.0, out=self.g_corr.view(B, 441, *self.grid[8]))
        self._corr_backward(self.g_corr, band)
        return (None if eng_window else self.g_c2a), self.g_c3a, self.g_c3b

    def _corr_backward(self, g_corr, band):
        from . import spatial_correlation_sampler_backend as correlation
        import ctypes as C
        h8, w8 = self.grid[8]
        if band is not None and band.cone_win is not None:
            L.check(L.lib().ufr_corr_backward_window(L.ptr(self._c3a), L.ptr(self._c3b), L.ptr(g_corr), L.ptr(self.g_c3a),
                                                     L.ptr(self.g_c3b), self.B, 256, h8, w8, 21, 2, L.ptr(band.cone_win), 8,
                                                     band.cone_hw[0] // 8, band.cone_hw[1] // 8, L.stream()),
                    "correlation backward (window)")
        else:
            p = correlation._params(1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
            L.check(L.lib().ufr_corr_backward(L.ptr(self._c3a), L.ptr(self._c3b), L.ptr(g_corr), L.ptr(self.g_c3a),
                                              L.ptr(self.g_c3b), L.UFR_F32, self.B, 256, h8, w8, C.byref(p), L.stream()),
                    "correlation backward")
        self.g_c3a.add_(self.g_c3a_redir)


class _EngineHead(torch.autograd.Function):
    """The engine keeps its activations in static buffers, so a forward is only differentiable until the NEXT forward of
    the same engine: `generation` counts forwards, backward refuses a stale one (two grad-mode forwards of one shape before
    a backward -- e.g. loss(flow(a), flow(b)) -- would otherwise silently differentiate the second call twice)."""

    @staticmethod
    def forward(ctx, c2a, c3a, c3b, engine, band):
        ctx.engine, ctx.band = engine, band
        engine.generation = ctx.generation = getattr(engine, "generation", 0) + 1
        # the engine's flow2 is a static buffer: hand autograd its own (2-channel, tiny) tensor -- or, inside a composition that
        # declared every consumer immediate (`_lib.static_handoff`), an alias of it
        ctx.static, ctx.static_grads = L.static_ok(), L.static_grads_ok()
        flow2 = engine.forward(c2a.contiguous(), c3a.contiguous(), c3b.contiguous() if c3b is not None else None, band)
        return flow2.detach() if ctx.static else flow2.clone()

    @staticmethod
    def backward(ctx, g_flow2):
        if ctx.engine.generation != ctx.generation:
            raise RuntimeError("FlowNetC head engine: another forward of this network (same batch and frame size) ran before this "
                               "backward; its activations are gone.  Call backward() before the next forward, or set "
                               "UFR_ENGINE=0 for interleaved forwards")
        g2a, g3a, g3b = ctx.engine.backward(g_flow2.contiguous(), ctx.band, fused_window=False)
        if ctx.static_grads:
            return g2a, g3a, g3b, None, None
        # static buffers as well: autograd (and a caller who retains .grad) gets its own copies
        return (g2a.clone() if g2a is not None else None, g3a.clone() if g3a is not None else None,
                g3b.clone() if g3b is not None else None, None, None)


def _weights_stamp(net):
    """(storage address, in-place version) of every parameter: the engine packs the weights once, so it must notice a
    `load_state_dict` / optimiser step on the module it was built from."""
    return tuple((p.data_ptr(), p._version) for p in net.parameters())


def get_engine(net, B: int, H: int, W: int, device) -> FlowNetCHeadEngine:
    """One engine per batch / frame size, cached on the module; rebuilt when the module's weights have changed since."""
    key = (int(B), int(H), int(W), str(torch.device(device)))
    cache = _engine_cache(net, "_ufr_head_engines")
    stamp = _weights_stamp(net)
    eng = cache.get(key)
    if eng is None or eng.weights_stamp != stamp:
        eng = cache[key] = FlowNetCHeadEngine(net, B, H, W, device)
        eng.weights_stamp = stamp
    return eng


def engine_head(net, c2a, c3a, c3b, band=None):
    """flow2 of FlowNetC's head on the native engine, as an autograd Function of the NCHW features."""
    B, _, h4, w4 = c2a.shape
    return _EngineHead.apply(c2a, c3a, c3b, get_engine(net, B, h4 * 4, w4 * 4, c2a.device), band)

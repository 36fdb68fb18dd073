"""Native RAFT refinement loop: the 12 iterations of models/raft/raft.py:189-228 -- cost-volume lookup, motion encoder,
SepConvGRU, flow head, and the mask head of the last iteration (models/raft/update.py:6-162) -- forward AND data gradient as an
explicit schedule of hand-written gfx950 kernels, one autograd Function around the whole loop.  Config C3 of BASELINE.json.

    every convolution (1x1, 3x3, 1x5, 5x1; 7x7 via gathered patches)   csrc/igemm.hip (bf16 split planes, six products)
    GRU gate arithmetic, flow patches, cat([out, flow])                csrc/raft_update.hip (chunk-major, streaming)
    lookup                                                              csrc/raft_altcorr_mfma.hip (alt_cuda_corr, one launch
                                                                        for all levels) or csrc/raft_corr.hip (all-pairs pyramid)
    delta_flow = Conv2d(256, 2, 3)                                      csrc/engine_small.hip (per-pixel GEMM + gather)

Layout.  A GRU half-step's two concatenations -- cat([h, x]) for the z / r gates and cat([r*h, x]) for q, x = cat([inp, motion])
(update.py:50-66) -- are ONE 12-chunk plane buffer [h 4 | motion 4 | r*h 4]: the gate convolution reads chunks 0-7, the
q convolution chunks 4-11 (its weights re-indexed once).  motion = cat([out 126, flow 2]) is exactly four chunks.

The context features `inp` are not in that buffer (round 5).  They are the SAME tensor in all 12 iterations (raft.py:176-180), and a
convolution is linear in its input channels: conv(W, [h | inp | motion]) = conv(W_h,m, [h | motion]) + conv(W_inp, inp).  The
second term of the four gate / candidate convolutions is computed ONCE per forward (float32 [M][256 | 128], `Cctx`) and added to
the pre-activations by the gate kernels (`addend` of the slab-reading forms; igemm's `add` epilogue otherwise): the 48 GRU
convolutions of a forward reduce over 256 instead of 384 channels.  Its adjoint is linear too: the pre-activation gradients of the
12 iterations are summed (the gate adjoint kernels add into `Azr` / `Aq`) and d / d inp is four transposed launches on the sums,
while the 48 per-iteration adjoint launches produce 256 instead of 384 columns.

What the loop's structure allows (raft.py:190: coords1 is detached at the top of every iteration): the flow head runs its
adjoint for the LAST iteration only, the whole flow branch of the motion encoder (convf1, convf2) has no adjoint at all, and the
mask head exists only in the last iteration.  Parameters are frozen (data gradients only).
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch
from ._lib import engine_cache as _engine_cache

from . import _lib as L
from . import igemm as ig
from .flownetc_engine import _pack_flow_head, _pack_flow_head_mfma

HC = 4                                  # chunks of the hidden state / context / motion features (128 channels each)


def _ptr(gs: ig.GradSum, chunk0: int) -> C.c_void_p:
    """Device pointer to chunk `chunk0` of a float32 [chunks][M][32] tensor."""
    return C.c_void_p(gs.t.data_ptr() + chunk0 * gs.M * 32 * 4)


class _Fork:
    """Work on the engine's second stream between a fork from and a join with the caller's stream.  Two ways to use it:
        with fork as on_side:            # joins at the end of the block
            with on_side: ...side work...
            ...main work...
        fork.start() as a context for the side work, fork.join() later (the backward's one-iteration lag).
    Without a side stream (UFR_RAFT_STREAMS=0) everything runs in program order on the caller's stream."""

    def __init__(self, side):
        self.side = side
        self.main = torch.cuda.current_stream(side.device) if side is not None else None
        self.used = False

    def start(self):
        import contextlib
        if self.side is None:
            return contextlib.nullcontext()
        self.side.wait_stream(self.main)
        self.used = True
        return torch.cuda.stream(self.side)

    def join(self):
        if self.used:
            self.main.wait_stream(self.side)
            self.used = False

    def __enter__(self):
        return self.start()

    def __exit__(self, *exc):
        self.join()


class RaftUpdateEngine:
    def __init__(self, net, B: int, H: int, W: int, device):
        if H % 8 or W % 8:
            raise ValueError("RAFT engine: frame sides must be multiples of 8")
        L.lib()
        self.net, self.B, self.H, self.W, self.dev = net, int(B), int(H), int(W), torch.device(device)
        self.h, self.w = H // 8, W // 8
        self.M = self.B * self.h * self.w
        self.iters = int(net.args.iters)
        self.radius, self.levels = int(net.args.corr_radius), int(net.args.corr_levels)
        self.generation = 0
        self._build()

    # ------------------------------------------------------------------------------------------------ set-up
    def _build(self):
        net, B, h, w, dev, IT = self.net, self.B, self.h, self.w, self.dev, self.iters
        ub, enc, gru = net.update_block, net.update_block.encoder, net.update_block.gru
        f32 = dict(dtype=torch.float32, device=dev)
        P = lambda chunks: ig.Planes(B, h, w, chunks, dev)
        G = lambda chunks: ig.GradSum(B, h, w, chunks, dev)
        cor_planes = self.levels * (2 * self.radius + 1) ** 2
        self.cor_planes, self.cor_chunks = cor_planes, ig.pad32(cor_planes) // 32
        # ---- activations: per iteration what an adjoint reads (ReLU masks, gate values), shared buffers for the rest
        self.corr_p = [P(self.cor_chunks) for _ in range(IT)]
        self.cor1 = [P(8) for _ in range(IT)]
        self.CF = [P(8) for _ in range(IT)]                           # [cor2 192 | flo2 64] = cat([cor, flo]) (update.py:118)
        self.P1 = [P(3 * HC) for _ in range(IT)] + [P(HC)]            # [h | motion | r*h]; the extra one holds the final h
        self.P2 = [P(3 * HC) for _ in range(IT)]
        self.INP = P(HC)                                              # the context features: one buffer, loaded once per forward
        # their share of the four gate / candidate pre-activations (raw sums, no bias), computed once per forward
        self.Cctx = {"zr1": G(2 * HC), "q1": G(HC), "zr2": G(2 * HC), "q2": G(HC)}
        self.ZR = [[G(2 * HC) for _ in range(IT)] for _ in range(2)]  # sigmoid(z) | sigmoid(r) of the two half-steps
        self.Q = [[G(HC) for _ in range(IT)] for _ in range(2)]       # tanh(q)
        self.fpat, self.flo1 = P(4), P(4)                             # flow branch: no adjoint, one buffer serves every iteration
        self.FH, self.MH = P(8), P(8)                                 # flow-head / mask-head hidden layers (last iteration's are kept)
        self.mask_f32 = G(18)
        self.corr = torch.zeros(B, cor_planes, h, w, **f32)
        self.flows = [torch.zeros(B, 2, h, w, **f32) for _ in range(IT)]
        self.delta = torch.zeros(B, 2, h, w, **f32)
        self.coords0 = torch.zeros(B, 2, h, w, **f32)
        ys, xs = torch.meshgrid(torch.arange(h, device=dev), torch.arange(w, device=dev), indexing="ij")
        self.coords0.copy_(torch.stack((xs, ys), dim=0).float()[None].expand(B, -1, -1, -1))     # utils/utils.py:80-83
        self.coords1 = torch.zeros(B, 2, h, w, **f32)
        self.coords_it = [torch.zeros(B, 2, h, w, **f32) for _ in range(IT + 1)]   # coords1 at the top of iteration it (the lookups' adjoints read them)
        self.flow_lr = torch.zeros(B, 2, h, w, **f32)
        self.up_mask = torch.zeros(B, 576, h, w, **f32)
        # ---- gradients
        # two running-sum buffers in the GRU buffer's own order [h | motion | r*h]: a half-step's adjoint launches ADD into
        # them in their epilogues (q^T: chunks 4-11 of the other buffer + its result -> this one; zr^T: chunks 0-7 in place),
        # so d / d h and the iteration's d / d motion need no separate add kernels (7 per iteration before)
        self.TA, self.TB, self.G_z = G(3 * HC), G(3 * HC), G(HC)
        # the pre-activation gradients summed over the iterations ([g_z | g_r] and g_q of the two half-steps: one arena, one fill),
        # their planes, and d / d inp
        self._acc_arena = torch.zeros(2 * (2 * HC + HC) * self.M * 32, **f32)
        self.Azr, self.Aq, off = [], [], 0
        for _ in range(2):
            for lst, ch in ((self.Azr, 2 * HC), (self.Aq, HC)):
                gs = G(1)
                gs.chunks, gs.t = ch, self._acc_arena[off:off + ch * self.M * 32].view(ch, self.M, 32)
                off += ch * self.M * 32
                lst.append(gs)
        self.gz_ctx = {"zr1": P(2 * HC), "q1": P(HC), "zr2": P(2 * HC), "q2": P(HC)}
        self.G_inp = G(HC)
        self.gzq, self.gzr, self.gz_mot = P(HC), P(2 * HC), P(HC)
        self.gz_cor2, self.gz_cor1 = P(6), P(8)
        self.G_corr = G(self.cor_chunks)
        self.g_corr = torch.zeros(B, cor_planes, h, w, **f32)
        self.G_fh, self.gz_fh = G(8), P(8)
        self.gz_mask, self.gz_mh = P(18), P(8)
        self.g_net0, self.g_inp = torch.zeros(B, 128, h, w, **f32), torch.zeros(B, 128, h, w, **f32)
        # ---- weights (frozen), in buffer channel order
        cw = lambda conv, pad, **kw: ig.conv_forward_weights(conv if torch.is_tensor(conv) else conv.weight, 1, pad, **kw)
        bw = lambda conv, pad: ig.conv_backward_weights(conv if torch.is_tensor(conv) else conv.weight, 1, pad)
        bias = lambda conv: conv.bias.detach().float().contiguous()
        wf1 = torch.zeros(128, 128, 1, 1, **f32)                      # convf1 over the gathered 7x7x2 patches: k = (ky*7 + kx)*2 + c
        wf1[:, :98, 0, 0] = enc.convf1.weight.detach().float().permute(0, 2, 3, 1).reshape(128, 98)
        w_zr, b_zr, w_q, w_zr_ctx, w_q_ctx = {}, {}, {}, {}, {}
        for tag in ("1", "2"):
            convz, convr, convq = (getattr(gru, f"conv{g}{tag}") for g in "zrq")
            wzr = torch.cat([convz.weight, convr.weight]).detach().float()                # [256, 384 = h | inp | motion, kh, kw]
            w_zr[tag] = torch.cat([wzr[:, :128], wzr[:, 256:]], 1).contiguous()           # buffer: [h | motion]
            w_zr_ctx[tag] = wzr[:, 128:256].contiguous()                                   # the context features' columns
            b_zr[tag] = torch.cat([convz.bias, convr.bias]).detach().float().contiguous()
            wq = convq.weight.detach().float()                                             # reference input: [r*h | inp | motion]
            w_q[tag] = torch.cat([wq[:, 256:], wq[:, :128]], 1).contiguous()              # buffer: [motion | r*h]
            w_q_ctx[tag] = wq[:, 128:256].contiguous()
        self._pads = {"1": (0, 2), "2": (2, 0)}
        plans = []

        def plan(key, wi, x, in_chunk0, ctx=None, **kw):
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            # ping-pong tiles with a deep split (>= 4 K tiles per slice) measured best on these 48 x 160 grids: 18.1 ms per
            # iteration against 18.4 (>= 8), 19.8 (>= 16), 18.4 with 64 x 128 tiles and 18.6 with single-stage 128 x 128 tiles
            kw.setdefault("variant", 6 if wi.Npad % 128 == 0 else 2)
            bm, target = (256, 256) if kw["variant"] == 6 else (128, 768)
            S = ig.splitk_for(self.M, wi.Npad, max(pk), 1, phase_ktiles=pk, bm=bm, target=target, min_ktiles=4)
            kw["variant"], S = ig.tuned(wi, self.M, kw, kw["variant"], S, rows=(h, w))
            # (capping the split because a second stream fills the chip anyway measured SLOWER: 16.02 ms -> 16.47 at <= 2 slices, 20.16
            # without split-K, gpurun r5_call11: a lone ping-pong workgroup per CU is latency-bound, short slices are the cure)
            if ctx is not None and not (kw.get("no_reduce") and S > 1):
                kw["add"] = ctx            # no slabs for the gate kernel to read: the context share rides in igemm's own epilogue
            plans.append((key, wi, x, in_chunk0, S, kw))

        W = dict(convc1=cw(enc.convc1, 0), convc2=cw(enc.convc2, 1), convf1=cw(wf1, 0), convf2=cw(enc.convf2, 1), conv=cw(enc.conv, 1),
                 fh1=cw(ub.flow_head.conv1, 1), mask1=cw(ub.mask[0], 1), mask2=cw(ub.mask[2], 0))
        Wb = dict(conv=bw(enc.conv.weight[:, :192], 1), convc2=bw(enc.convc2, 1), convc1=bw(enc.convc1, 0), fh1=bw(ub.flow_head.conv1, 1),
                  mask1=bw(ub.mask[0], 1), mask2=bw(ub.mask[2], 0))
        for tag in ("1", "2"):
            W["zr" + tag], W["q" + tag] = cw(w_zr[tag], self._pads[tag]), cw(w_q[tag], self._pads[tag])
            Wb["zr" + tag], Wb["q" + tag] = bw(w_zr[tag], self._pads[tag]), bw(w_q[tag], self._pads[tag])
            W["ctx_zr" + tag], W["ctx_q" + tag] = cw(w_zr_ctx[tag], self._pads[tag]), cw(w_q_ctx[tag], self._pads[tag])
            Wb["ctx_zr" + tag], Wb["ctx_q" + tag] = bw(w_zr_ctx[tag], self._pads[tag]), bw(w_q_ctx[tag], self._pads[tag])
        relu, lin = dict(slope=0.0), dict(slope=1.0)
        b_q = {tag: bias(getattr(gru, "convq" + tag)) for tag in ("1", "2")}
        b_conv = self._conv_bias = bias(enc.conv)
        self._gate_bias = {**{"zr" + t: b_zr[t] for t in ("1", "2")}, **{"q" + t: b_q[t] for t in ("1", "2")}}
        fuse = os.environ.get("UFR_RAFT_FUSE_REDUCE", "1") != "0"
        # once per forward / backward: the context features' share of the gate convolutions and its adjoint (a chain of adds)
        for i, name in enumerate(("zr1", "q1", "zr2", "q2")):
            plan(("ctx_" + name,), W["ctx_" + name], self.INP, 0, out_f32=self.Cctx[name])
            plan(("ctx_" + name + "^T",), Wb["ctx_" + name], self.gz_ctx[name], 0, out_f32=self.G_inp,
                 **(dict(add=self.G_inp) if i else {}))
        for it in range(IT):
            P1, P2 = self.P1[it], self.P2[it]
            plan(("convc1", it), W["convc1"], self.corr_p[it], 0, out_planes=self.cor1[it], bias=bias(enc.convc1), **relu)
            plan(("convc2", it), W["convc2"], self.cor1[it], 0, out_planes=self.CF[it], out_chunk0=0, bias=bias(enc.convc2), **relu)
            plan(("convf1", it), W["convf1"], self.fpat, 0, out_planes=self.flo1, bias=bias(enc.convf1), **relu)
            plan(("convf2", it), W["convf2"], self.flo1, 0, out_planes=self.CF[it], out_chunk0=6, bias=bias(enc.convf2), **relu)
            plan(("conv", it), W["conv"], self.CF[it], 0, out_planes=P1, out_chunk0=HC, bias=b_conv, no_reduce=fuse, **relu)
            for half, (tag, buf, nxt) in enumerate((("1", P1, P2), ("2", P2, self.P1[it + 1]))):
                # the gate / candidate convolutions leave their split-K slabs to the gate arithmetic (no reduce launch in between)
                plan(("zr" + tag, it), W["zr" + tag], buf, 0, ctx=self.Cctx["zr" + tag], out_f32=self.ZR[half][it], bias=b_zr[tag],
                     no_reduce=fuse, **lin)
                plan(("q" + tag, it), W["q" + tag], buf, HC, ctx=self.Cctx["q" + tag], out_f32=self.Q[half][it], bias=b_q[tag],
                     no_reduce=fuse, **lin)
                # adjoints: d / d [motion | r*h] of q, d / d [h | motion] of the gates
                src, dst = (self.TA, self.TB) if tag == "2" else (self.TB, self.TA)    # the backward walks half-step 2 first
                plan(("q" + tag + "^T", it), Wb["q" + tag], self.gzq, 0, add=src, add_chunk0=HC, out_f32=dst, out_f32_chunk0=HC)
                plan(("zr" + tag + "^T", it), Wb["zr" + tag], self.gzr, 0, add=dst, add_chunk0=0, out_f32=dst, out_f32_chunk0=0)
            plan(("fh1", it), W["fh1"], self.P1[it + 1], 0, out_planes=self.FH, bias=bias(ub.flow_head.conv1), **relu)
            # motion encoder adjoint (the correlation branch only): masks are the ReLU outputs of that iteration
            plan(("conv^T", it), Wb["conv"], self.gz_mot, 0, mask=self.CF[it], out_planes=self.gz_cor2, **relu)
            plan(("convc2^T", it), Wb["convc2"], self.gz_cor2, 0, mask=self.cor1[it], out_planes=self.gz_cor1, **relu)
            plan(("convc1^T", it), Wb["convc1"], self.gz_cor1, 0, out_f32=self.G_corr)
        last = self.P1[IT]
        plan(("mask1",), W["mask1"], last, 0, out_planes=self.MH, bias=bias(ub.mask[0]), **relu)
        plan(("mask2",), W["mask2"], self.MH, 0, out_f32=self.mask_f32, bias=bias(ub.mask[2]), **lin)
        plan(("mask2^T",), Wb["mask2"], self.gz_mask, 0, mask=self.MH, out_planes=self.gz_mh, **relu)
        plan(("mask1^T",), Wb["mask1"], self.gz_mh, 0, out_f32=self.TA, out_f32_chunk0=0)
        plan(("fh1^T",), Wb["fh1"], self.gz_fh, 0, add=self.TA, add_chunk0=0, out_f32=self.TA, out_f32_chunk0=0)
        # Two streams (round 5).  At one pair (48 x 160 = 7,680 rows) no launch of the update block fills the 256 CUs, and two parts
        # of an iteration do not depend on the chain that carries the hidden state:
        #   forward   the flow branch of the motion encoder (flow patches, convf1, convf2) next to lookup -> convc1 -> convc2;
        #   backward  the motion encoder's adjoint and the lookup's adjoint of iteration i (conv^T .. convc1^T, alt_corr / lookup
        #             backward: they end in the feature maps' gradients, coords are detached, raft.py:190) next to the GRU adjoint
        #             of iteration i - 1.
        # The side stream's launches split K into their own workspace.  Inside a HIP-graph capture the fork / join events become
        # graph edges.  UFR_RAFT_STREAMS=0: everything on the caller's stream (the A/B switch).
        self._side_keys = {"convf1", "convf2", "conv^T", "convc2^T", "convc1^T"}
        side = lambda key: key[0] in self._side_keys
        need = max([S * self.M * wi.Npad for key, wi, _, _, S, _ in plans if S > 1 and not side(key)] + [1])
        need_side = max([S * self.M * wi.Npad for key, wi, _, _, S, _ in plans if S > 1 and side(key)] + [1])
        self.ws, self.ws_side = torch.empty(need, **f32), torch.empty(need_side, **f32)
        self.launch, self._wi = {}, {}
        for key, wi, x, c0, S, kw in plans:
            ws = (self.ws_side if side(key) else self.ws) if S > 1 else None
            self.launch[key] = ig.make_launch(wi, x, c0, (h, w), (h, w), splitk=S, ws=ws, **kw)
            self._wi[key] = wi
        self._two_streams = os.environ.get("UFR_RAFT_STREAMS", "1") != "0"
        self._side_stream = torch.cuda.Stream(device=dev) if self._two_streams else None
        fh2 = ub.flow_head.conv2
        self.fh2_w, self.fh2_wm = _pack_flow_head(fh2.weight), _pack_flow_head_mfma(fh2.weight)
        self.fh2_b = fh2.bias.detach().float().contiguous()

    @staticmethod
    def _slices(launch) -> int:
        """Slabs a single-phase split-K launch writes: ufr_igemm cuts the K tiles into ceil(KT / splitk)-sized slices."""
        d = launch.desc
        kt = d.phase[0].ntaps * d.KC
        per = -(-kt // d.splitk)
        return -(-kt // per)

    def launch_table(self):
        rows = []
        for key, launch in self.launch.items():
            d = launch.desc
            rows.append(("_".join(str(v) for v in key), launch, self._wi[key].flops(d.B * d.Hr * d.Wr) / 1e9))
        return rows

    def _fork(self):
        return _Fork(self._side_stream)

    # ------------------------------------------------------------------------------------------------ lookups
    def _alt_levels(self, f2s, grads=None):
        from .flownets.raft_corr import _levels_struct
        return _levels_struct(f2s, grads)

    def _lookup_forward(self, src, coords):
        lib, B, h, w = L.lib(), self.B, self.h, self.w
        if src["alt"] and src.get("planes") is not None:
            src["planes"].forward(coords, self.radius, src["scale"], out=self.corr)
        elif src["alt"]:
            lv = self._alt_levels(src["f2"])
            L.check(lib.ufr_altcorr_pyramid_forward(L.ptr(src["f1"]), C.byref(lv), L.ptr(coords), L.ptr(self.corr), B, h, w,
                                                    src["f1"].shape[3], self.radius, src["scale"], L.stream()), "alt_corr forward")
        else:
            from .flownets.raft_corr import _pyramid_struct
            pyr = _pyramid_struct(src["vols"])
            L.check(lib.ufr_corr_lookup_forward(C.byref(pyr), L.ptr(coords), L.ptr(self.corr), B, h, w, self.radius, L.stream()),
                    "corr lookup forward")

    def _lookup_backward(self, src, coords, first):
        lib, B, h, w = L.lib(), self.B, self.h, self.w
        if src["alt"] and not src.get("dense"):
            lv = self._alt_levels(src["f2"], src["g_f2"])
            # (the cost volume's gradient is read as convc1^T left it, chunk-major: no NCHW copy in between)
            L.check(lib.ufr_altcorr_pyramid_backward_cm(L.ptr(src["f1"]), C.byref(lv), L.ptr(coords), L.ptr(self.G_corr.t), L.ptr(src["g_f1"]),
                                                     L.ptr(src["ws"]), B, h, w, src["f1"].shape[3], self.radius, src["scale"],
                                                     0 if first else 1, L.stream()), "alt_corr backward")
        else:
            from .flownets.raft_corr import _pyramid_struct
            # (alt_cuda_corr with the dense adjoint: the window adjoints go into gradient volumes too, raft_corr.alt_dense_adjoint; the
            #  adjoint kernel reads the levels' SHAPES from the first list only)
            pyr = _pyramid_struct(src["g_vols"] if src["alt"] else src["vols"], src["g_vols"])   # accumulates (+=): the buffers were zeroed
            L.check(lib.ufr_corr_lookup_backward(C.byref(pyr), L.ptr(coords), L.ptr(self.g_corr), B, h, w, self.radius, L.stream()),
                    "corr lookup backward")

    # ------------------------------------------------------------------------------------------------ the schedule
    @torch.no_grad()
    def forward(self, net0: torch.Tensor, inp: torch.Tensor, src: dict, flow_init: torch.Tensor | None = None):
        """net0 = tanh(context[:128]), inp = relu(context[128:]) [B,128,H/8,W/8]; src = the lookup's operands.
        -> (coords1 - coords0 [B,2,H/8,W/8], 0.25 * mask [B,576,H/8,W/8]) of the last iteration (raft.py:213-233, test_mode)."""
        lib, st, B, h, w, M, IT = L.lib(), L.stream, self.B, self.h, self.w, self.M, self.iters
        self.generation += 1
        self._src = src
        self.P1[0].load_nchw(net0.contiguous(), 0)
        self.INP.load_nchw(inp.contiguous(), 0)                       # the context features are the same in every iteration:
        for name in ("zr1", "q1", "zr2", "q2"):                       # their share of the gate convolutions, once
            self.launch[("ctx_" + name,)]()
        self.coords1.copy_(self.coords0)
        if flow_init is not None:
            self.coords1.add_(flow_init)
        self._coords = self.coords_it
        n_coords = self.coords1.numel()
        L.check(lib.ufr_raft_coords_step(L.ptr(self.coords1), None, L.ptr(self.coords0), L.ptr(self.coords_it[0]), L.ptr(self.flows[0]),
                                         n_coords, st()), "coords step")
        for it in range(IT):
            coords = self.coords_it[it]                               # (the adjoint of this iteration's lookup reads them)
            with self._fork() as on_side:                             # the flow branch: no adjoint, nothing of the lookup in it
                with on_side:
                    L.check(lib.ufr_raft_flow_patches(L.ptr(self.flows[it]), L.ptr(self.fpat.t), self.fpat.plane_stride, 0, B, h, w, st()),
                            "flow patches")
                    self.launch[("convf1", it)]()
                    self.launch[("convf2", it)]()
                self._lookup_forward(src, coords)
                self.corr_p[it].load_nchw(self.corr, 0)
                self.launch[("convc1", it)]()
                self.launch[("convc2", it)]()
            self.launch[("conv", it)]()
            P1, P2 = self.P1[it], self.P2[it]
            lc = self.launch[("conv", it)]
            if lc.desc.no_reduce:                                    # its slabs -> both GRU buffers' motion chunks (+ the flow channels)
                L.check(lib.ufr_raft_motion_finish_slabs(L.ptr(self.ws), self._slices(lc), lc.desc.Npad, lc.desc.N, L.ptr(self._conv_bias), 0.0,
                                                         L.ptr(P1.t), P1.plane_stride, L.ptr(P2.t), P2.plane_stride, HC,
                                                         L.ptr(self.flows[it]), B, h, w, st()), "motion finish (slabs)")
            else:
                L.check(lib.ufr_raft_motion_finish(L.ptr(P1.t), P1.plane_stride, L.ptr(P2.t), P2.plane_stride, HC, L.ptr(self.flows[it]),
                                                   B, h, w, st()), "motion finish")
            for half, (tag, buf, nxt) in enumerate((("1", P1, P2), ("2", P2, self.P1[it + 1]))):
                ZR, Q = self.ZR[half][it], self.Q[half][it]
                lz = self.launch[("zr" + tag, it)]
                lz()
                if lz.desc.no_reduce:
                    L.check(lib.ufr_gru_gates_cm_forward_slabs(L.ptr(self.ws), self._slices(lz), lz.desc.Npad, L.ptr(self._gate_bias["zr" + tag]),
                                                               L.ptr(self.Cctx["zr" + tag].t), L.ptr(ZR.t), L.ptr(buf.t), buf.plane_stride, 0,
                                                               L.ptr(buf.t), buf.plane_stride, 2 * HC, M, HC, st()), "gru gates forward (slabs)")
                else:
                    L.check(lib.ufr_gru_gates_cm_forward(L.ptr(ZR.t), L.ptr(buf.t), buf.plane_stride, 0, L.ptr(buf.t), buf.plane_stride,
                                                         2 * HC, M, HC, st()), "gru gates forward")
                lq = self.launch[("q" + tag, it)]
                lq()
                if lq.desc.no_reduce:
                    L.check(lib.ufr_gru_blend_cm_forward_slabs(L.ptr(self.ws), self._slices(lq), lq.desc.Npad, L.ptr(self._gate_bias["q" + tag]),
                                                               L.ptr(self.Cctx["q" + tag].t), L.ptr(Q.t), L.ptr(ZR.t), L.ptr(buf.t),
                                                               buf.plane_stride, 0, L.ptr(nxt.t), nxt.plane_stride, 0, M, HC, st()),
                            "gru blend forward (slabs)")
                else:
                    L.check(lib.ufr_gru_blend_cm_forward(L.ptr(Q.t), L.ptr(ZR.t), L.ptr(buf.t), buf.plane_stride, 0, L.ptr(nxt.t),
                                                         nxt.plane_stride, 0, M, HC, st()), "gru blend forward")
            self.launch[("fh1", it)]()
            L.check(lib.ufr_flow_head_planes_forward_mfma(L.ptr(self.FH.t), self.FH.plane_stride, 0, 8, L.ptr(self.fh2_wm), self.fh2_wm.shape[0],
                                                          L.ptr(self.fh2_b),
                                                          L.ptr(self.delta), B, h, w, st()), "delta_flow")
            # coords1 += delta_flow, the next iteration's copy and flow = coords1 - coords0 in one kernel (it == IT - 1: into the spare slot)
            nxt_flow = self.flows[it + 1] if it + 1 < IT else self.flow_lr
            L.check(lib.ufr_raft_coords_step(L.ptr(self.coords1), L.ptr(self.delta), L.ptr(self.coords0), L.ptr(self.coords_it[it + 1]),
                                             L.ptr(nxt_flow), n_coords, st()), "coords step")
        self.launch[("mask1",)]()
        self.launch[("mask2",)]()
        self.mask_f32.to_nchw(576, 0, scale=0.25, slope=1.0, out=self.up_mask)
        return self.flow_lr, self.up_mask                             # (flow_lr = coords1 - coords0: the last coords step wrote it)

    @torch.no_grad()
    def backward(self, g_flow: torch.Tensor, g_mask: torch.Tensor | None):
        """(d loss / d flow_lr, d loss / d up_mask) -> (d / d net0, d / d inp) NCHW; the lookup operands' gradients are added into
        src['g_f1'] / src['g_f2'] (alt_corr) or src['g_vols'] (all-pairs)."""
        lib, st, B, h, w, M, IT = L.lib(), L.stream, self.B, self.h, self.w, self.M, self.iters
        src = self._src
        # ---- last iteration: mask head and flow head -> d / d h
        if g_mask is not None:
            self.gz_mask.load_nchw(g_mask.contiguous(), 0, scale=0.25)
            self.launch[("mask2^T",)]()
            self.launch[("mask1^T",)]()
        else:
            self.TA.t[:HC].zero_()
        L.check(lib.ufr_flow_head_planes_backward(L.ptr(g_flow.contiguous()), L.ptr(self.fh2_w), self.fh2_w.shape[0], L.ptr(self.G_fh.t),
                                                  self.G_fh.chunks, 0, 8, B, h, w, 0,
                                                  st()), "delta_flow backward")
        L.check(lib.ufr_grad_finalize(L.ptr(self.G_fh.t), 0, L.ptr(self.FH.t), 0, L.ptr(self.gz_fh.t), self.gz_fh.plane_stride, 0, M, 8,
                                      0.0, st()), "flow head finalize")
        self.launch[("fh1^T",)]()
        self.TA.t[HC:].zero_()                                        # d / d motion; the r*h slots start at zero
        self.TB.t[2 * HC:].zero_()
        self._acc_arena.zero_()                                       # the pre-activation gradients' sums over the iterations
        pending = None                                                # the side stream's work of the previous iteration
        for it in range(IT - 1, -1, -1):
            # cur = chunks 0-3 of `gin` (d / d the half-step's output h), prev = chunks 0-3 of `gout` (d / d its input h)
            for half, (tag, buf, gin, gout) in ((1, ("2", self.P2[it], self.TA, self.TB)), (0, ("1", self.P1[it], self.TB, self.TA))):
                ZR, Q = self.ZR[half][it], self.Q[half][it]
                L.check(lib.ufr_gru_blend_cm_backward(L.ptr(Q.t), L.ptr(ZR.t), L.ptr(buf.t), buf.plane_stride, 0, L.ptr(gin.t),
                                                      L.ptr(self.gzq.t), self.gzq.plane_stride, 0, L.ptr(self.G_z.t), L.ptr(gout.t), M,
                                                      HC, L.ptr(self.Aq[half].t), st()), "gru blend backward")
                self.launch[("q" + tag + "^T", it)]()                 # gout[motion | r*h] = gin[...] + d / d [motion | r*h]
                L.check(lib.ufr_gru_gates_cm_backward(L.ptr(ZR.t), L.ptr(buf.t), buf.plane_stride, 0, L.ptr(self.G_z.t), _ptr(gout, 2 * HC),
                                                      L.ptr(self.gzr.t), self.gzr.plane_stride, 0, L.ptr(gout.t), M, HC, 1,
                                                      L.ptr(self.Azr[half].t), st()),
                        "gru gates backward")                         # (consumes the r*h slot: zeros for the next adder)
                self.launch[("zr" + tag + "^T", it)]()                # gout[h | motion] += d / d [h | motion]
            # motion features -> ReLU' -> conv^T (correlation branch) -> convc2^T -> convc1^T -> the lookup's adjoint: this chain
            # ends in the feature maps' gradients, the GRU adjoint of the next (earlier) iteration does not wait for it
            if pending is not None:
                pending.join()                                         # (its buffers -- gz_mot .. g_corr -- are about to be rewritten)
            # (the next iteration's motion gradient starts from zero: the finalize kernel leaves zeros behind its read)
            L.check(lib.ufr_grad_finalize_consume(_ptr(self.TA, HC), C.c_void_p(self.P1[it].t.data_ptr() + HC * M * 32 * 2), L.ptr(self.gz_mot.t),
                                                  self.gz_mot.plane_stride, M, HC, 0.0, st()), "motion finalize")
            pending = self._fork()
            with pending.start():
                for name in ("conv^T", "convc2^T", "convc1^T"):
                    self.launch[(name, it)]()
                if not src["alt"] or src.get("dense"):                 # (alt_corr's on-the-fly adjoint reads the chunk-major sum itself)
                    self.G_corr.to_nchw(self.cor_planes, 0, slope=1.0, out=self.g_corr)
                self._lookup_backward(src, self._coords[it], first=(it == IT - 1))
        if pending is not None:
            pending.join()
        self.TA.to_nchw(128, 0, slope=1.0, out=self.g_net0)          # (an even number of half-steps: TA holds d / d net0)
        # d / d inp: the adjoint of the context share, once, on the pre-activation gradients summed over the iterations
        for half, tag in enumerate(("1", "2")):
            for name, acc in (("zr" + tag, self.Azr[half]), ("q" + tag, self.Aq[half])):
                gz = self.gz_ctx[name]
                L.check(lib.ufr_grad_finalize(L.ptr(acc.t), 0, None, 0, L.ptr(gz.t), gz.plane_stride, 0, M, acc.chunks, 1.0, st()),
                        "context share: gradient sums -> planes")
        for name in ("zr1", "q1", "zr2", "q2"):
            self.launch[("ctx_" + name + "^T",)]()
        self.G_inp.to_nchw(128, 0, slope=1.0, out=self.g_inp)
        return self.g_net0, self.g_inp


class _RaftRefine(torch.autograd.Function):
    """The refinement loop as one Function of (net0, inp, lookup operands); static buffers: see flownetc_engine._EngineHead."""

    @staticmethod
    def forward(ctx, net0, inp, engine, alt, scale, f1, *rest):
        if alt:
            src = dict(alt=True, f1=f1, f2=list(rest), scale=float(scale))
            from .flownets.raft_corr import AltCorrPlanes
            if AltCorrPlanes.served(f1, rest, engine.radius):      # the maps as split planes, once per forward: the 12 lookups read them
                src["planes"] = AltCorrPlanes(f1, list(rest))
        else:
            src = dict(alt=False, vols=list(rest))
        flow_lr, up_mask = engine.forward(net0, inp, src)
        ctx.engine, ctx.src, ctx.generation = engine, src, engine.generation
        return flow_lr.clone(), up_mask.clone()

    @staticmethod
    def backward(ctx, g_flow, g_mask):
        eng, src = ctx.engine, ctx.src
        if eng.generation != ctx.generation:
            raise RuntimeError("RAFT engine: another forward of this network (same batch and frame size) ran before this backward; its "
                               "activations are gone.  Call backward() before the next forward, or set UFR_ENGINE=0")
        from .flownets.raft_corr import alt_dense_adjoint, alt_dense_adjoint_served, alt_dense_volumes
        if src["alt"] and alt_dense_adjoint_served(src["f1"], src["f2"], eng.radius):
            src["dense"], src["g_vols"] = True, alt_dense_volumes(src["f1"], src["f2"])
        elif src["alt"]:
            f1 = src["f1"]
            src["g_f1"], src["g_f2"] = torch.empty_like(f1), [torch.empty_like(f) for f in src["f2"]]
            nbytes = L.lib().ufr_altcorr_pyramid_workspace_bytes(eng.B, eng.h, eng.w, f1.shape[3], eng.radius, len(src["f2"]))
            src["ws"] = torch.empty(nbytes, dtype=torch.uint8, device=f1.device)
        else:
            src["g_vols"] = [torch.zeros_like(v) for v in src["vols"]]
        g_net0, g_inp = eng.backward(g_flow, g_mask)
        if src["alt"] and src.get("dense"):
            g_f1, g_f2 = alt_dense_adjoint(src["f1"], src["f2"], src["g_vols"], src["scale"])
            src["g_vols"] = None                                       # (313 MB per pair: released with this backward)
            return (g_net0.clone(), g_inp.clone(), None, None, None, g_f1, *g_f2)
        if src["alt"]:
            return (g_net0.clone(), g_inp.clone(), None, None, None, src["g_f1"], *src["g_f2"])
        return (g_net0.clone(), g_inp.clone(), None, None, None, None, *src["g_vols"])


def get_engine(net, B: int, H: int, W: int, device) -> RaftUpdateEngine:
    from .flownetc_engine import _weights_stamp
    a = net.args            # the launch schedule bakes these in: a changed `args.iters` on a live model must rebuild it
    key = (int(B), int(H), int(W), str(torch.device(device)), int(a.iters), int(a.corr_radius), int(a.corr_levels), ig._PRODUCTS[-1])
    cache = _engine_cache(net, "_ufr_head_engines")
    stamp = _weights_stamp(net)
    eng = cache.get(key)
    if eng is None or eng.weights_stamp != stamp:
        eng = cache[key] = RaftUpdateEngine(net, B, H, W, device)
        eng.weights_stamp = stamp
    return eng


def refine(net, net0, inp, corr_fn, H: int, W: int):
    """(flow at 1/8 resolution, up_mask) of the 12 iterations on the engine; corr_fn = the AlternateCorrBlock / CorrBlock the
    reference's loop would call (its operands are taken, its per-iteration autograd Functions are not used)."""
    eng = get_engine(net, net0.shape[0], H, W, net0.device)
    from .flownets.raft_corr import AlternateCorrBlock
    if isinstance(corr_fn, AlternateCorrBlock):
        dim = corr_fn._f1.shape[3]
        return _RaftRefine.apply(net0, inp, eng, True, 1.0 / math.sqrt(dim), corr_fn._f1, *corr_fn._f2)
    return _RaftRefine.apply(net0, inp, eng, False, 1.0, None, *[v.contiguous() for v in corr_fn.corr_pyramid])

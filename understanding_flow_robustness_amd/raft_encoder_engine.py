"""Native RAFT feature / context encoder: models/raft/extractor.py:142-215 (BasicEncoder: 7x7/2 stem, three residual stages of
two ResidualBlocks (:5-78), 1x1 head) forward AND data gradient as an explicit schedule of hand-written gfx950 kernels.

    every convolution (7x7/2 over packed planes, 3x3 stride 1 / 2, 1x1 stride 1 / 2)   csrc/igemm.hip -> float32 [chunks][M][32]
    InstanceNorm2d (fnet) / folded BatchNorm2d (cnet), ReLU, residual add              csrc/raft_norm.hip -> activation planes

fnet normalises with the statistics of each image (no affine, biased variance, eps 1e-5): statistics kernel + apply kernel
behind every convolution, and the adjoint's two per-image means (sum g, sum g * xhat) the same way.  cnet's BatchNorm2d runs in
eval mode (the attack freezes the network): scale and shift are folded into the convolution's weights and bias once.
Parameters are frozen (data gradients only).
"""
from __future__ import annotations

import torch
from ._lib import engine_cache as _engine_cache

from . import _lib as L
from . import igemm as ig

EPS = 1e-5


def _fold(conv, norm):
    """(weight, bias) of conv followed by an eval-mode BatchNorm2d; plain (weight, bias) for anything else."""
    w, b = conv.weight.detach().float(), conv.bias.detach().float()
    if isinstance(norm, torch.nn.BatchNorm2d):
        s = norm.weight.detach().float() / torch.sqrt(norm.running_var.detach().float() + norm.eps)
        return w * s.view(-1, 1, 1, 1), (b - norm.running_mean.detach().float()) * s + norm.bias.detach().float()
    return w, b


def _ds_backward_weights(weight: torch.Tensor) -> ig.WeightImage:
    """Data gradient of Conv2d(C, N, 1, stride 2): gx[2y, 2x] = W^T gy[y, x], zero elsewhere -- ONE phase of the transposed form
    (the caller zero-fills the output first; rows = the gy grid, out_s = 2)."""
    N, Cn = weight.shape[:2]
    mat = weight.detach().float().reshape(N, Cn).t().reshape(Cn, 1, N).contiguous()
    planes, offsets, n, npad, KC, cr = ig._pack([mat], weight.device)
    return ig.WeightImage(planes, offsets, [(0, 0, [(0, 0)])], n, npad, KC, dict(in_s=1, out_s=2), cr)


class _Unit:
    """conv -> norm -> ReLU (-> + residual -> ReLU): X = conv(inp) float32, out planes; backward: G_out -> gz -> conv^T."""

    def __init__(self, eng, conv, norm, inp: ig.Planes, stride: int, out: ig.Planes, res: ig.Planes | None, relu1: bool, relu2: bool,
                 packed=None):
        self.eng, self.inp, self.out, self.res, self.relu1, self.relu2 = eng, inp, out, res, relu1, relu2
        n, (ho, wo) = out.B, (out.H, out.W)
        w, b = _fold(conv, norm)
        self.cout = w.shape[0]
        self.X = ig.GradSum(n, ho, wo, ig.pad32(self.cout) // 32, eng.dev)
        self.stats = torch.zeros(n * self.X.chunks * 32 * 2, dtype=torch.float32, device=eng.dev) if eng.inorm else None
        self.sums = torch.zeros_like(self.stats) if eng.inorm else None
        self.gz = ig.Planes(n, ho, wo, self.X.chunks, eng.dev)
        k, pad = w.shape[-1], (w.shape[-1] - 1) // 2
        self.packed = packed
        if packed is not None:                                   # the 7x7 / 2 stem over the packed planes of the raw frames
            self.wi, self.wi_b = ig.conv1_packed_weights(w), ig.conv1_packed_backward_weights(w)
            self.fwd = eng.launch(self.wi, packed, (ho, wo), (ho, wo), out_f32=self.X, bias=b.contiguous(), slope=1.0, variant=2)
            self.G_p = ig.GradSum(n, ho + 3, wo + 2, 1, eng.dev)
            self.bwd = eng.launch(self.wi_b, self.gz, (ho + 3, wo + 2), (ho + 3, wo + 2), out_f32=self.G_p)
        else:
            self.wi = ig.conv_forward_weights(w, stride, pad)
            self.fwd = eng.launch(self.wi, inp, (ho, wo), (ho, wo), out_f32=self.X, bias=b.contiguous(), slope=1.0)
            self.stride, self.k = stride, k
            self.wi_b = _ds_backward_weights(w) if (k == 1 and stride == 2) else ig.conv_backward_weights(w, stride, pad)
            self.bwd = None                                      # built by `adjoint_into` once the destination is known

    def adjoint_into(self, G_in: ig.GradSum, add: ig.GradSum | None = None):
        """conv^T writes d / d inp into G_in (+ add)."""
        kw = dict(add=add) if add is not None else {}
        self.bwd = self.eng.launch(self.wi_b, self.gz, (self.out.H, self.out.W), (self.inp.H, self.inp.W), out_f32=G_in, **kw)

    def forward(self):
        lib, st, X, e = L.lib(), L.stream(), self.X, self.eng
        self.fwd()
        HW = self.out.H * self.out.W
        res = self.res
        if self.stats is not None:             # InstanceNorm: partial sums, then the apply kernel finishes the statistics itself
            L.check(lib.ufr_cm_norm_stats_apply(L.ptr(X.t), L.ptr(self.stats), L.ptr(e.ws_d), EPS,
                                                L.ptr(res.t) if res is not None else None, res.plane_stride if res is not None else 0, 0,
                                                L.ptr(self.out.t), self.out.plane_stride, 0, HW, X.B, X.chunks, int(self.relu1),
                                                int(self.relu2), st), "norm stats + apply")
            return
        L.check(lib.ufr_cm_norm_apply(L.ptr(X.t), L.ptr(self.stats) if self.stats is not None else None,
                                      L.ptr(res.t) if res is not None else None, res.plane_stride if res is not None else 0, 0,
                                      L.ptr(self.out.t), self.out.plane_stride, 0, HW, X.B, X.chunks, int(self.relu1), int(self.relu2), st),
                "norm apply")

    def backward(self, G_out: ig.GradSum):
        """d / d out (float32) -> gradient planes of the convolution's output -> conv^T."""
        lib, st, X, e = L.lib(), L.stream(), self.X, self.eng
        HW = self.out.H * self.out.W
        mask = self.out if self.relu2 else None
        L.check(lib.ufr_cm_norm_backward(L.ptr(X.t), L.ptr(G_out.t), L.ptr(mask.t) if mask is not None else None, 0,
                                         L.ptr(self.stats) if self.stats is not None else None,
                                         L.ptr(self.sums) if self.sums is not None else None, L.ptr(e.ws_d), L.ptr(self.gz.t),
                                         self.gz.plane_stride, 0, HW, X.B, X.chunks, int(self.relu1), st), "norm backward")
        self.bwd()


class RaftEncoderEngine:
    def __init__(self, enc, n: int, H: int, W: int, device):
        if H % 8 or W % 8:
            raise ValueError("RAFT encoder engine: frame sides must be multiples of 8")
        L.lib()
        self.enc, self.n, self.H, self.W, self.dev = enc, int(n), int(H), int(W), torch.device(device)
        self.inorm = enc.norm_fn == "instance"
        if enc.norm_fn not in ("instance", "batch"):
            raise NotImplementedError("RAFT encoder engine: instance (fnet) or batch (cnet, eval) normalisation")
        self.generation = 0
        self._plans = []
        self._build()

    def launch(self, wi, x, rows, out_hw, **kw):
        """Deferred igemm launch (one split-K workspace for the whole encoder is sized at the end of _build)."""
        kw.setdefault("variant", 6 if wi.Npad % 128 == 0 else 7)
        holder = _Deferred()
        self._plans.append((holder, wi, x, rows, out_hw, kw))
        return holder

    def _build(self):
        enc, n, dev = self.enc, self.n, self.dev
        H2, W2 = self.H // 2, self.W // 2
        P = lambda s, c: ig.Planes(n, self.H // s, self.W // s, ig.pad32(c) // 32, dev)
        G = lambda s, c: ig.GradSum(n, self.H // s, self.W // s, ig.pad32(c) // 32, dev)
        self.packed = ig.Planes(n, H2 + 3, W2 + 2, 1, dev)
        self.zero_mean = torch.zeros(3, dtype=torch.float64, device=dev)
        a0 = P(2, 64)
        self.stem = _Unit(self, enc.conv1, enc.norm1, None, 2, a0, None, True, False, packed=self.packed)
        self.blocks = []                                         # (unit1, unit2, unit_ds | None, x planes, out planes, G_x, G_y1, G_s)
        x, cin, s = a0, 64, 2
        for layer, cout, stride in ((enc.layer1, 64, 1), (enc.layer2, 96, 2), (enc.layer3, 128, 2)):
            for bi, rb in enumerate(layer):
                st_ = stride if bi == 0 else 1
                so = s * st_
                y1, out = P(so, cout), P(so, cout)
                u1 = _Unit(self, rb.conv1, rb.norm1, x, st_, y1, None, True, False)
                uds, xd = None, x
                if rb.downsample is not None:
                    xd = P(so, cout)
                    uds = _Unit(self, rb.downsample[0], rb.norm3, x, st_, xd, None, False, False)
                u2 = _Unit(self, rb.conv2, rb.norm2, y1, 1, out, xd, True, True)
                G_x, G_y1, G_s = G(s, cin), G(so, cout), G(so, cout)
                u2.adjoint_into(G_y1)
                if uds is None:
                    u1.adjoint_into(G_x, add=G_s)                # d / d x = conv1^T(...) + the skip connection's share
                else:
                    uds.adjoint_into(G_x)                        # even positions of a zero-filled G_x ...
                    u1.adjoint_into(G_x, add=G_x)                # ... then conv1^T adds the rest (all four phases)
                self.blocks.append(dict(u1=u1, u2=u2, uds=uds, x=x, out=out, G_x=G_x, G_y1=G_y1, G_s=G_s))
                x, cin, s = out, cout, so
        # head: Conv2d(128, output_dim, 1) with bias, no normalisation
        w2 = enc.conv2.weight.detach().float()
        self.cout = w2.shape[0]
        self.X_out = G(8, self.cout)
        self.head = self.launch(ig.conv_forward_weights(w2, 1, 0), x, (x.H, x.W), (x.H, x.W), out_f32=self.X_out,
                                bias=enc.conv2.bias.detach().float().contiguous(), slope=1.0)
        self.gz_out = P(8, self.cout)
        self.G_last = G(8, 128)
        self.head_b = self.launch(ig.conv_backward_weights(w2, 1, 0), self.gz_out, (x.H, x.W), (x.H, x.W), out_f32=self.G_last)
        self.out_nchw = torch.zeros(n, self.cout, x.H, x.W, dtype=torch.float32, device=dev)
        self.g_image = torch.zeros(n, 3, self.H, self.W, dtype=torch.float32, device=dev)
        self.G_a0 = self.blocks[0]["G_x"]
        # ---- one split-K workspace, one float64 workspace for the statistics
        sized = []
        for holder, wi, xin, rows, out_hw, kw in self._plans:
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            v = kw["variant"]
            bm, target = (256, 256) if v in (6, 7) else (128, 768)
            S = ig.splitk_for(n * rows[0] * rows[1], wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target, min_ktiles=4)
            kw["variant"], S = ig.tuned(wi, n * rows[0] * rows[1], kw, v, S, rows=rows)
            sized.append(S)
        need = max([len(p[1].phases) * S * n * p[3][0] * p[3][1] * p[1].Npad for p, S in zip(self._plans, sized) if S > 1] + [1])
        self.ws = torch.empty(need, dtype=torch.float32, device=dev)
        for (holder, wi, xin, rows, out_hw, kw), S in zip(self._plans, sized):
            holder.launch = ig.make_launch(wi, xin, 0, rows, out_hw, splitk=S, ws=self.ws if S > 1 else None, **kw)
            holder.wi = wi
        self.ws_d = torch.empty(L.lib().ufr_cm_norm_workspace_doubles(H2 * W2, n, 4), dtype=torch.float64, device=dev)

    def launch_table(self):
        rows = []
        for i, (holder, wi, xin, r, o, kw) in enumerate(self._plans):
            d = holder.launch.desc
            rows.append((f"enc_{i}", holder.launch, wi.flops(d.B * d.Hr * d.Wr) / 1e9))
        return rows

    @torch.no_grad()
    def forward(self, images: torch.Tensor) -> torch.Tensor:
        """Normalised frames [n, 3, H, W] -> features [n, output_dim, H/8, W/8] (NCHW float32, a static buffer)."""
        L.require_hip(images, "images")
        if tuple(images.shape) != (self.n, 3, self.H, self.W) or images.dtype != torch.float32:
            raise RuntimeError("RAFT encoder engine: frames of another shape")
        self.generation += 1
        pk = self.packed
        L.check(L.lib().ufr_conv1_pack_planes(L.ptr(images.contiguous()), None, L.ptr(pk.t), pk.plane_stride, self.n, 0, self.H, self.W,
                                              L.ptr(self.zero_mean), L.stream()), "stem pack")
        self.stem.forward()
        for blk in self.blocks:
            blk["u1"].forward()
            if blk["uds"] is not None:
                blk["uds"].forward()
            blk["u2"].forward()
        self.head()
        return self.X_out.to_nchw(self.cout, 0, slope=1.0, out=self.out_nchw)

    @torch.no_grad()
    def backward(self, g_out: torch.Tensor) -> torch.Tensor:
        """d loss / d features -> d loss / d frames [n, 3, H, W] (a static buffer)."""
        lib, st = L.lib(), L.stream
        self.gz_out.load_nchw(g_out.contiguous(), 0)
        self.head_b()
        G_out = self.G_last
        for blk in reversed(self.blocks):
            u1, u2, uds = blk["u1"], blk["u2"], blk["uds"]
            u2.backward(G_out)                                   # -> G_y1
            elems = G_out.t.numel()
            L.check(lib.ufr_cm_masked_copy(L.ptr(G_out.t), L.ptr(blk["out"].t), 0, L.ptr(blk["G_s"].t), elems, st()), "skip gradient")
            if uds is not None:
                blk["G_x"].t.zero_()
                uds.backward(blk["G_s"])                         # no ReLU behind norm3: G_s is already masked by the block's output
            u1.backward(blk["G_y1"])
            G_out = blk["G_x"]
        self.stem.backward(G_out)
        L.check(lib.ufr_conv1_unpack_grad(L.ptr(self.stem.G_p.t), L.ptr(self.g_image), self.n, self.H, self.W, st()), "stem unpack")
        return self.g_image


class _Deferred:
    """A prepared igemm launch whose split-K workspace is assigned once every launch of the engine is known."""
    launch = None

    def __call__(self):
        self.launch()


class _RaftEncoder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, engine):
        out = engine.forward(images).clone()
        ctx.engine, ctx.generation = engine, engine.generation
        return out

    @staticmethod
    def backward(ctx, g):
        if ctx.engine.generation != ctx.generation:
            raise RuntimeError("RAFT encoder engine: another forward of this encoder (same batch and frame size) ran before this "
                               "backward; its activations are gone.  Call backward() before the next forward, or set UFR_ENGINE=0")
        return ctx.engine.backward(g).clone(), None


def encode(enc, images: torch.Tensor) -> torch.Tensor:
    """BasicEncoder.forward on the native engine (one engine per encoder, batch and frame size, cached on the module)."""
    from .flownetc_engine import _weights_stamp
    n, _, H, W = images.shape
    key = (int(n), int(H), int(W), str(images.device), ig._PRODUCTS[-1])       # (the launches bake the products of their arithmetic in)
    cache = _engine_cache(enc, "_ufr_encoder_engines")
    stamp = _weights_stamp(enc) + tuple((b.data_ptr(), b._version) for b in enc.buffers())
    eng = cache.get(key)
    if eng is None or eng.weights_stamp != stamp:
        eng = cache[key] = RaftEncoderEngine(enc, n, H, W, images.device)
        eng.weights_stamp = stamp
    return _RaftEncoder.apply(images, eng)

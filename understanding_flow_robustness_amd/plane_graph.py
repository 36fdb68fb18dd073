"""A small schedule builder for the convolutional sub-networks that have no bespoke engine: FlowNet2's FlowNetSD and
FlowNetFusion and the conv1-3 prefixes of its FlowNetS stacks (models/flownet2/FlowNetSD.py:12-126, FlowNetFusion.py:12-71,
FlowNetS.py:15-104) -- every layer forward AND data gradient on the hand-written gfx950 kernels, activations resident in
the plane layout between layers.

    conv / deconv / inter_conv            csrc/igemm.hip (bf16 split planes, six products; bias + LeakyReLU or linear epilogue)
    predict_flow*, upsampled_flow*        csrc/engine_small.hip
    7x7 stride-2 stem on 12 channels      pixel-unshuffle pack (csrc/plane_layout.hip) + a 16-tap stride-1 launch over 48 channels

A `torch.cat` of the reference is a chunk offset into one buffer (a segment list says where each member sits; weights are
re-indexed once).  Adjoint: every buffer has a float32 gradient sum of the same layout; the first operator adjoint to reach a
segment writes its share, the later ones ADD theirs (igemm epilogue `add` = its own output, the 2-channel kernels with
accumulate; `_plan_first_writers` -- round 4 zero-filled every sum at the start of a backward), then LeakyReLU' and the split
into gradient planes happen once per produced segment.  Parameters are frozen.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import igemm as ig
from .flownetc_engine import _pack_flow_head, _pack_flow_head_mfma


class _Buf:
    def __init__(self, B, H, W, chunks, dev):
        self.planes = ig.Planes(B, H, W, chunks, dev)
        self.grad = ig.GradSum(B, H, W, chunks, dev)
        self.H, self.W, self.chunks = H, W, chunks


def remap_in(weight: torch.Tensor, segments, cbuf: int, dim: int = 1) -> torch.Tensor:
    """Move the reference's input channels to their buffer positions: segments = [(first reference channel, count, first buffer
    channel)]; everything else of the `cbuf` buffer channels gets zero weights."""
    shape = list(weight.shape)
    shape[dim] = cbuf
    w = torch.zeros(shape, dtype=torch.float32, device=weight.device)
    src = weight.detach().float()
    for r0, n, b0 in segments:
        idx_w = [slice(None)] * w.dim()
        idx_s = [slice(None)] * w.dim()
        idx_w[dim], idx_s[dim] = slice(b0, b0 + n), slice(r0, r0 + n)
        w[tuple(idx_w)] = src[tuple(idx_s)]
    return w


class PlaneGraph:
    def __init__(self, B: int, device):
        L.lib()
        self.B, self.dev = int(B), torch.device(device)
        self.bufs: dict[str, _Buf] = {}
        self.ops = []                      # forward order
        self._plans = []
        self.flows: dict[str, torch.Tensor] = {}
        self.g_flows: dict[str, torch.Tensor] = {}
        self.inputs, self.outputs, self.tensor_outputs = [], [], []
        self.generation = 0

    # ------------------------------------------------------------------------------------------------ building
    def buffer(self, name, H, W, chunks):
        self.bufs[name] = _Buf(self.B, H, W, chunks, self.dev)
        return self.bufs[name]

    def _launch(self, wi, x, in_chunk0, rows, out_hw, **kw):
        holder = _Deferred()
        kw.setdefault("variant", 6 if wi.Npad % 128 == 0 else 7)
        self._plans.append((holder, wi, x, in_chunk0, rows, out_hw, kw))
        return holder

    def input(self, buf, channels, chunk0=0):
        """An NCHW float32 tensor [B, channels, H, W] loaded into `buf` at chunk0; its gradient is returned by backward()."""
        b = self.bufs[buf]
        self.inputs.append(dict(buf=b, C=channels, chunk0=chunk0, g=torch.zeros(self.B, channels, b.H, b.W, dtype=torch.float32, device=self.dev)))

    def input_packed12(self, buf, channels):
        """A [B, channels, 2H, 2W] tensor entering as its 2x2 pixel-unshuffle (channel (c*2 + p)*2 + q = x[c, 2y + p, 2x + q]) for a
        7x7 stride-2 stem run as a 16-tap stride-1 launch (`conv7x7s2_unshuffled`)."""
        b = self.bufs[buf]
        self.inputs.append(dict(buf=b, C=channels, chunk0=0, packed=True,
                                g=torch.zeros(self.B, channels, 2 * b.H, 2 * b.W, dtype=torch.float32, device=self.dev)))

    def conv(self, weight, bias, src, dst, stride=1, slope=ig.LEAKY, in_segments=None, taps_unshuffled=False):
        """dst segment = act(conv(src segment)).  src = (buffer, first chunk, chunks), dst = (buffer, first chunk).
        in_segments: where the reference's input channels sit inside the src chunks (default: contiguous from channel 0)."""
        sb, s0, sk = self.bufs[src[0]], src[1], src[2]
        db, d0 = self.bufs[dst[0]], dst[1]
        w = weight.detach().float()
        if taps_unshuffled:               # Conv2d(C, N, 7, 2, 3) over the 2x2-unshuffled input: 4x4 taps a, b in [-2, 1], channel (c, p, q)
            N, Cn = w.shape[:2]
            wu = torch.zeros(N, 4 * Cn, 4, 4, dtype=torch.float32, device=w.device)
            for ky in range(7):
                a, p = (ky - 3) // 2, (ky - 3) % 2
                for kx in range(7):
                    bb, q = (kx - 3) // 2, (kx - 3) % 2
                    wu[:, (torch.arange(Cn, device=w.device) * 2 + p) * 2 + q, a + 2, bb + 2] = w[:, :, ky, kx]
            w, stride, pad = wu, 1, 2     # taps (a, b) = (ky' - 2, kx' - 2): "padding 2" of a 4x4 kernel
        else:
            pad = (w.shape[-1] - 1) // 2
        if in_segments is not None or w.shape[1] != sk * 32:
            w = remap_in(w, in_segments or [(0, w.shape[1], 0)], sk * 32)
        N = w.shape[0]
        nch = ig.pad32(N) // 32
        wi = ig.conv_forward_weights(w, stride, pad)
        wib = ig.conv_backward_weights(w, stride, pad)
        rows = (db.H, db.W)
        gz = ig.Planes(self.B, db.H, db.W, nch, self.dev)
        fwd = self._launch(wi, sb.planes, s0, rows, rows, out_planes=db.planes, out_chunk0=d0, bias=bias.detach().float().contiguous(), slope=slope)
        bwd = self._launch(wib, gz, 0, rows, (sb.H, sb.W), add=sb.grad, add_chunk0=s0, out_f32=sb.grad, out_f32_chunk0=s0)
        self.ops.append(dict(kind="conv", fwd=fwd, bwd=bwd, gz=gz, db=db, d0=d0, nch=nch, slope=slope, wi=wi, wib=wib, sb=sb, s0=s0, sk=sk))

    def deconv(self, weight, bias, src, dst, slope=ig.LEAKY, in_segments=None):
        """ConvTranspose2d(Cin, Cout, 4, 2, 1) + bias + LeakyReLU (submodules.py:75-82)."""
        sb, s0, sk = self.bufs[src[0]], src[1], src[2]
        db, d0 = self.bufs[dst[0]], dst[1]
        w = weight.detach().float()                         # [Cin, Cout, 4, 4]
        if in_segments is not None or w.shape[0] != sk * 32:
            w = remap_in(w, in_segments or [(0, w.shape[0], 0)], sk * 32, dim=0)
        N = w.shape[1]
        nch = ig.pad32(N) // 32
        wi, wib = ig.deconv_forward_weights(w, 1), ig.deconv_backward_weights(w, 1)
        gz = ig.Planes(self.B, db.H, db.W, nch, self.dev)
        fwd = self._launch(wi, sb.planes, s0, (sb.H, sb.W), (db.H, db.W), out_planes=db.planes, out_chunk0=d0,
                           bias=bias.detach().float().contiguous(), slope=slope)
        bwd = self._launch(wib, gz, 0, (sb.H, sb.W), (sb.H, sb.W), add=sb.grad, add_chunk0=s0, out_f32=sb.grad, out_f32_chunk0=s0)
        self.ops.append(dict(kind="conv", fwd=fwd, bwd=bwd, gz=gz, db=db, d0=d0, nch=nch, slope=slope, wi=wi, wib=wib, sb=sb, s0=s0, sk=sk))

    def predict_flow(self, conv, src, name, in_segments=None):
        """Conv2d(C, 2, 3, 1, 1) on a buffer's chunks -> flow `name` [B, 2, H, W] (NCHW float32)."""
        sb, s0, sk = self.bufs[src[0]], src[1], src[2]
        w = conv.weight.detach().float()
        if in_segments is not None or w.shape[1] != sk * 32:
            w = remap_in(w, in_segments or [(0, w.shape[1], 0)], sk * 32)
        self.flows[name] = torch.zeros(self.B, 2, sb.H, sb.W, dtype=torch.float32, device=self.dev)
        self.g_flows[name] = torch.zeros_like(self.flows[name])
        self.ops.append(dict(kind="pf", sb=sb, s0=s0, sk=sk, wm=_pack_flow_head_mfma(w), wb=_pack_flow_head(w),
                             b=conv.bias.detach().float().contiguous(), name=name))

    def up_flow(self, deconv, flow, dst):
        """ConvTranspose2d(2, 2, 4, 2, 1) of flow `flow` into lanes 0-1 of one chunk of `dst` = (buffer, chunk)."""
        db, chunk = self.bufs[dst[0]], dst[1]
        self.ops.append(dict(kind="up", flow=flow, db=db, chunk=chunk, w=deconv.weight.detach().float().contiguous(),
                             b=deconv.bias.detach().float().contiguous() if deconv.bias is not None else None))

    def output(self, flow):
        self.outputs.append(flow)

    def tensor_output(self, buf, channels, chunk0=0):
        """A buffer segment handed out as an NCHW float32 tensor (a feature map another engine consumes); its gradient comes
        back as NCHW and is added to the buffer's gradient sum."""
        b = self.bufs[buf]
        self.tensor_outputs.append(dict(buf=b, C=channels, chunk0=chunk0,
                                        t=torch.zeros(self.B, channels, b.H, b.W, dtype=torch.float32, device=self.dev)))

    def _fuse_single_reader_segments(self):
        """A produced segment whose FIRST reader is a convolution over exactly that segment (conv3 -> conv3_1, conv0 -> conv1,
        convK_1 -> conv(K+1) in front of a concatenation, ...) needs no `grad_finalize` launch: that convolution's transposed launch
        -- the last contribution to arrive in the backward -- writes LeakyReLU'(segment) x (its result + what the other readers
        left in the float32 sum) straight into the producer's gradient planes (igemm epilogue `add` + `mask` + `out_planes`)."""
        plans = {id(p[0]): p for p in self._plans}
        overlap = lambda a0, an, b0, bn: a0 < b0 + bn and b0 < a0 + an
        for P in self.ops:
            if P["kind"] != "conv":
                continue
            db, d0, nch = P["db"], P["d0"], P["nch"]
            # (an "up" op PRODUCES its chunk of db in the forward and only reads db.grad in the backward: it is a writer, below)
            readers = [Q for Q in self.ops if Q["kind"] in ("conv", "pf") and Q["sb"] is db and overlap(Q["s0"], Q["sk"], d0, nch)]
            # the fusion rests on single assignment: every reader comes AFTER its producer in the forward order and nothing else
            # writes the segment -- a reused segment would hand P's gradient planes to the wrong launch without any error
            at = self.ops.index(P)
            if any(self.ops.index(Q) <= at for Q in readers):
                raise RuntimeError(f"PlaneGraph: a reader of {P.get('name', 'a convolution')}'s output segment runs before it (segment reused?)")
            for Q in self.ops:
                if Q is not P and ((Q["kind"] == "conv" and Q["db"] is db and overlap(Q["d0"], Q["nch"], d0, nch)) or
                                   (Q["kind"] == "up" and Q["db"] is db and overlap(Q["chunk"], 1, d0, nch))):
                    raise RuntimeError(f"PlaneGraph: two producers write chunks [{d0}, {d0 + nch}) of one buffer")
            if any(t["buf"] is db and overlap(t["chunk0"], -(-t["C"] // 32), d0, nch) for t in self.tensor_outputs):
                continue
            # the reader that comes FIRST in the forward order runs LAST in the backward: when it is a convolution over exactly
            # this segment, its transposed launch completes the sum (epilogue `add` = what the other readers left) and finalises
            if not readers or readers[0]["kind"] != "conv" or (readers[0]["s0"], readers[0]["sk"]) != (d0, nch):
                continue
            kw = plans[id(readers[0]["bwd"])][6]
            kw.pop("out_f32", None); kw.pop("out_f32_chunk0", None)
            if len(readers) == 1:
                kw.pop("add", None); kw.pop("add_chunk0", None)
            kw.update(out_planes=P["gz"], out_chunk0=0, slope=float(P["slope"]))
            if P["slope"] != 1.0:
                kw.update(mask=db.planes, mask_chunk0=d0)
            P["finalized"] = True

    def _plan_first_writers(self):
        """Which adjoint launch WRITES a gradient segment first.  The backward used to start with one fill of every gradient sum of
        the graph (~90 MB per FlowNet2 sub-network at 448 x 1024, 11 fills = 0.2 ms of a C5 iteration); instead the first launch that
        reaches a chunk range writes `=` (the igemm epilogue without `add`, the 2-channel kernels without accumulate) and the later
        ones `+=`.  Walks the backward in its own order with the set of chunks written so far per buffer; a range that is partly
        written when a launch reaches it, or read before anything wrote it (a produced segment nobody reads), stays on the list of
        zero fills.  Returns that list (tensor views)."""
        written = {id(b): set() for b in self.bufs.values()}
        zero = {id(b): set() for b in self.bufs.values()}
        plans = {id(p[0]): p for p in self._plans}

        def write(buf, c0, n) -> bool:
            R, w = set(range(c0, c0 + n)), written[id(buf)]
            fresh = not (R & w)
            if not fresh:
                zero[id(buf)] |= R - w
            w |= R
            return fresh

        def read(buf, c0, n):
            missing = set(range(c0, c0 + n)) - written[id(buf)]
            zero[id(buf)] |= missing
            written[id(buf)] |= missing

        flows_written, zero_flows = set(self.outputs), []
        for spec in self.tensor_outputs:
            spec["overwrite"] = write(spec["buf"], spec["chunk0"], spec["C"] // 32)
        for op in reversed(self.ops):
            if op["kind"] == "conv":
                if not op.get("finalized"):
                    read(op["db"], op["d0"], op["nch"])
                kw = plans[id(op["bwd"])][6]
                if "out_f32" in kw:                               # (else: a fused finalize, which writes its producer's gradient planes)
                    if write(op["sb"], op["s0"], op["sk"]):
                        kw.pop("add", None); kw.pop("add_chunk0", None)
                elif "add" in kw:
                    read(op["sb"], op["s0"], op["sk"])
            elif op["kind"] == "pf":
                if op["name"] not in flows_written:               # a flow nobody consumes: its gradient is zero
                    zero_flows.append(op["name"])
                    flows_written.add(op["name"])
                op["acc"] = 0 if write(op["sb"], op["s0"], op["sk"]) else 1
            else:
                read(op["db"], op["chunk"], 1)
                op["acc"] = 1 if op["flow"] in flows_written else 0
                flows_written.add(op["flow"])
        for spec in self.inputs:
            read(spec["buf"], spec["chunk0"], (4 * spec["C"] + 31) // 32 if spec.get("packed") else -(-spec["C"] // 32))
        fills = []
        for b in self.bufs.values():
            chunks = sorted(zero[id(b)])
            while chunks:                                          # maximal runs of chunks: one fill each
                c0 = c1 = chunks.pop(0)
                while chunks and chunks[0] == c1 + 1:
                    c1 = chunks.pop(0)
                fills.append((b, c0, c1 + 1))
        return fills, zero_flows

    def build(self):
        B = self.B
        if __import__("os").environ.get("UFR_GRAPH_FUSE_FINALIZE", "1") != "0":
            self._fuse_single_reader_segments()
        if __import__("os").environ.get("UFR_GRAPH_ZERO_ARENA", "0") == "1":          # round 4's form: one fill of everything
            fills, zero_flows = None, None
            for spec in self.tensor_outputs:
                spec["overwrite"] = False
            for op in self.ops:
                op["acc"] = 1
        else:
            fills, zero_flows = self._plan_first_writers()
        # every gradient sum and flow gradient in one arena: a backward starts with ONE fill instead of one per buffer
        r64 = lambda n: (n + 63) // 64 * 64                   # keep every member 256-byte aligned
        sizes = [b.grad.t.numel() for b in self.bufs.values()] + [r64(g.numel()) for g in self.g_flows.values()]
        self._zero_arena = torch.zeros(sum(sizes), dtype=torch.float32, device=self.dev)
        off = 0
        for b in self.bufs.values():
            n = b.grad.t.numel()
            b.grad.t = self._zero_arena[off:off + n].view_as(b.grad.t)
            off += n
        for k in list(self.g_flows):
            n = self.g_flows[k].numel()
            self.g_flows[k] = self._zero_arena[off:off + n].view_as(self.g_flows[k])
            off += r64(n)
        # what a backward still has to zero before its first launch (None: everything, as one fill)
        self._zero_list = None if fills is None else ([b.grad.t[c0:c1] for b, c0, c1 in fills] + [self.g_flows[k] for k in zero_flows])
        sized = []
        for holder, wi, x, c0, rows, out_hw, kw in self._plans:
            pk = [len(t) * wi.KC for _, _, t in wi.phases]
            bm, target = (256, 256) if kw["variant"] in (6, 7) else (128, 768)
            S = ig.splitk_for(B * rows[0] * rows[1], wi.Npad, max(pk), len(wi.phases), phase_ktiles=pk, bm=bm, target=target, min_ktiles=4)
            kw["variant"], S = ig.tuned(wi, B * rows[0] * rows[1], kw, kw["variant"], S, rows=rows)
            sized.append(S)
        need = max([len(p[1].phases) * S * B * p[4][0] * p[4][1] * p[1].Npad for p, S in zip(self._plans, sized) if S > 1] + [1])
        self.ws = torch.empty(need, dtype=torch.float32, device=self.dev)
        for (holder, wi, x, c0, rows, out_hw, kw), S in zip(self._plans, sized):
            holder.launch = ig.make_launch(wi, x, c0, rows, out_hw, splitk=S, ws=self.ws if S > 1 else None, **kw)
            holder.wi = wi
        return self

    def launch_table(self):
        rows = []
        for i, (holder, wi, x, c0, r, o, kw) in enumerate(self._plans):
            d = holder.launch.desc
            rows.append((f"launch_{i}", holder.launch, wi.flops(d.B * d.Hr * d.Wr) / 1e9))
        return rows

    # ------------------------------------------------------------------------------------------------ running
    @torch.no_grad()
    def forward(self, *tensors):
        lib, st = L.lib(), L.stream
        self.generation += 1
        for spec, x in zip(self.inputs, tensors):
            L.require_hip(x, "input")
            b = spec["buf"]
            if spec.get("packed"):
                L.check(lib.ufr_unshuffle_pack_planes(L.ptr(x.contiguous()), L.ptr(b.planes.t), b.planes.plane_stride, self.B, spec["C"], b.H, b.W,
                                                      st()), "unshuffle pack")
            else:
                b.planes.load_nchw(x.contiguous(), spec["chunk0"])
        for op in self.ops:
            if op["kind"] == "conv":
                op["fwd"]()
            elif op["kind"] == "pf":
                sb = op["sb"]
                L.check(lib.ufr_flow_head_planes_forward_mfma(L.ptr(sb.planes.t), sb.planes.plane_stride, op["s0"], op["sk"], L.ptr(op["wm"]),
                                                              op["wm"].shape[0], L.ptr(op["b"]), L.ptr(self.flows[op["name"]]), self.B, sb.H, sb.W, st()),
                        "predict_flow forward")
            else:
                f, db = self.flows[op["flow"]], op["db"]
                L.check(lib.ufr_flow_up_planes_forward(L.ptr(f), L.ptr(op["w"]), L.ptr(op["b"]) if op["b"] is not None else None,
                                                       L.ptr(db.planes.t), db.planes.plane_stride, op["chunk"], self.B, f.shape[2], f.shape[3],
                                                       st()), "upsampled_flow forward")
        outs = [self.flows[n] for n in self.outputs]
        for spec in self.tensor_outputs:
            outs.append(spec["buf"].planes.to_nchw(spec["C"], spec["chunk0"], out=spec["t"]))
        return outs

    @torch.no_grad()
    def backward(self, *grads):
        lib, st = L.lib(), L.stream
        if self._zero_list is None:
            self._zero_arena.zero_()
        else:
            for t in self._zero_list:                                   # (what no adjoint launch overwrites first: `_plan_first_writers`)
                t.zero_()
        for name, g in zip(self.outputs, grads):
            self.g_flows[name].copy_(g)
        for spec, g in zip(self.tensor_outputs, grads[len(self.outputs):]):
            b, C_, c0 = spec["buf"], spec["C"], spec["chunk0"]          # NCHW -> chunk-major float32, the first share of the buffer's sum
            n = C_ // 32
            if C_ % 32:
                raise NotImplementedError("tensor outputs are whole chunks")
            src = g.view(self.B, n, 32, b.H, b.W).permute(1, 0, 3, 4, 2)
            if spec["overwrite"]:
                b.grad.t[c0:c0 + n].view(n, self.B, b.H, b.W, 32).copy_(src)
            else:
                b.grad.t[c0:c0 + n].add_(src.reshape(n, -1, 32))
        for op in reversed(self.ops):
            if op["kind"] == "conv":
                db, gz = op["db"], op["gz"]
                if not op.get("finalized"):                      # (else its single reader's transposed launch wrote gz already)
                    mask = db.planes if op["slope"] != 1.0 else None
                    L.check(lib.ufr_grad_finalize(L.ptr(db.grad.t), op["d0"], L.ptr(mask.t) if mask is not None else None, op["d0"], L.ptr(gz.t),
                                                  gz.plane_stride, 0, db.grad.M, op["nch"], float(op["slope"]), st()), "gradient finalize")
                op["bwd"]()
            elif op["kind"] == "pf":
                sb = op["sb"]
                L.check(lib.ufr_flow_head_planes_backward(L.ptr(self.g_flows[op["name"]]), L.ptr(op["wb"]), op["wb"].shape[0], L.ptr(sb.grad.t),
                                                          sb.grad.chunks, op["s0"], op["sk"],
                                                          self.B, sb.H, sb.W, op["acc"], st()), "predict_flow backward")
            else:
                gf, db = self.g_flows[op["flow"]], op["db"]
                L.check(lib.ufr_flow_up_planes_backward(L.ptr(db.grad.t), op["chunk"], L.ptr(op["w"]), L.ptr(gf), self.B, gf.shape[2], gf.shape[3],
                                                        op["acc"], st()), "upsampled_flow backward")      # += : an output flow already holds its gradient
        outs = []
        for spec in self.inputs:
            b = spec["buf"]
            if spec.get("packed"):
                L.check(lib.ufr_unshuffle_unpack_grad(L.ptr(b.grad.t), L.ptr(spec["g"]), self.B, spec["C"], b.H, b.W, st()), "unshuffle unpack")
            else:
                b.grad.to_nchw(spec["C"], spec["chunk0"], slope=1.0, out=spec["g"])
            outs.append(spec["g"])
        return outs


class _Deferred:
    launch = None

    def __call__(self):
        self.launch()


class _GraphFunction(torch.autograd.Function):
    """A PlaneGraph as an autograd Function of its NCHW inputs (static buffers: a forward is differentiable until the next one)."""

    @staticmethod
    def forward(ctx, graph, *inputs):
        ctx.static = L.static_ok()            # (`_lib.static_handoff`: every consumer reads at once -> aliases instead of clones)
        ctx.static_grads = L.static_grads_ok()   # (... and the inputs have one consumer each: their gradients may be aliases too)
        outs = [o.detach() if ctx.static else o.clone() for o in graph.forward(*inputs)]
        ctx.graph, ctx.generation = graph, graph.generation
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        if ctx.graph.generation != ctx.generation:
            raise RuntimeError("native sub-network: another forward (same batch and frame size) ran before this backward; its "
                               "activations are gone.  Call backward() before the next forward, or set UFR_ENGINE=0")
        gs = ctx.graph.backward(*[g.contiguous() for g in grads])
        return (None, *[g if ctx.static_grads else g.clone() for g in gs])


def run(graph: PlaneGraph, *inputs):
    return _GraphFunction.apply(graph, *inputs)


def graph_for(module, key, build):
    """One PlaneGraph per (batch, frame size, device), cached on the module; rebuilt when its weights changed since."""
    from .flownetc_engine import _weights_stamp
    cache = L.engine_cache(module, "_ufr_plane_graphs")
    stamp = _weights_stamp(module)
    g = cache.get(key)
    if g is None or g.weights_stamp != stamp:
        g = cache[key] = build()
        g.weights_stamp = stamp
    return g


def native_ok(module, x) -> bool:
    """plane_graph.py serves the attack's configuration: frozen parameters, eval mode, HIP float32, sides multiples of 64;
    a forward that is refused says so once (`_lib.engine_gate`)."""
    return L.engine_gate(module, x, 64)


def stem_graph(net, n, H, W, cin, dev):
    """conv1 (7x7 / 2 over the 2x2-unshuffled input), conv2, conv3 (5x5 / 2) of FlowNetC / FlowNetS (models/FlowNetC.py:100-119,
    models/flownet2/FlowNetS.py:15-60) -> conv2 and conv3 as NCHW tensors; `net` has conv1 / conv2 / conv3 = Sequential(Conv2d, LeakyReLU)."""
    g = PlaneGraph(n, dev)
    g.buffer("pin", H // 2, W // 2, (4 * cin + 31) // 32)
    g.buffer("c1", H // 2, W // 2, 2)
    g.buffer("c2", H // 4, W // 4, 4)
    g.buffer("c3", H // 8, W // 8, 8)
    g.input_packed12("pin", cin)
    c1, c2, c3 = net.conv1[0], net.conv2[0], net.conv3[0]
    g.conv(c1.weight, c1.bias, ("pin", 0, (4 * cin + 31) // 32), ("c1", 0), taps_unshuffled=True)
    g.conv(c2.weight, c2.bias, ("c1", 0, 2), ("c2", 0), stride=2)
    g.conv(c3.weight, c3.bias, ("c2", 0, 4), ("c3", 0), stride=2)
    g.tensor_output("c2", 128)
    g.tensor_output("c3", 256)
    return g.build()

"""RAFT (raft-things configuration: BasicEncoder x2, 4-level all-pairs correlation, SepConvGRU update
block, convex x8 upsampling, 12 iterations) on the gfx950 cost-volume kernels.

Behavioural mirror of models/raft/raft.py:26-233, update.py, extractor.py:1-215 with the layer names
of the reference (so `raft-things.pth` loads unchanged).  Differences, all on purpose:
  * the per-iteration lookup is ONE fused kernel (flownets/raft_corr.py) instead of 4 grid_samples;
  * `alternate_corr=True` is differentiable (the reference calls alt_cuda_corr's raw forward);
  * float32 by DEFAULT, whatever `args.mixed_precision` says: the reference's registry sets that flag for every non-"adv" RAFT
    (models/utils_model.py:51) and then runs both encoders and the update block under fp16 autocast on a GPU
    (models/raft/raft.py:140,168,195); the 1e-4 EPE / patch gates of this build are defined against the float32 path
    ("RAFT precision"), so the flag ALONE changes nothing here.  Reduced precision is an explicit opt-in on top of
    the flag: `args.mixed_precision` AND the environment's `UFR_RAFT_PRECISION=bf16` make the native engines' convolutions (both
    encoders, the update block; the correlation stays float32 like the reference's `.float()` operands, corr.py:128-129,
    raft.py:147-148) compute ONE bf16 product per float32 product (csrc/igemm.hip `products = 1`: bf16-rounded operands, float32
    accumulation -- what a bfloat16 autocast computes, 1/6 of the matrix work); `UFR_RAFT_PRECISION=bf16x3` the three leading
    products (~16 significand bits, 1/2 of the work).  `RAFT.products()` says which form a forward will take.  The torch
    fallback path (training mode, parameters that want gradients) ignores the switch and stays float32.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from ..band_conv import FlowHead as _FlowConv
from ..band_conv import conv_relu
from .raft_corr import AlternateCorrBlock, CorrBlock


def _norm(kind, planes, groups=None):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups or planes // 8, num_channels=planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    return nn.Sequential()


class ResidualBlock(nn.Module):
    """extractor.py:5-78."""

    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.non_linearity = nn.ReLU(inplace=True)
        self.norm1, self.norm2 = _norm(norm_fn, planes), _norm(norm_fn, planes)
        self.downsample = None
        if stride != 1:
            self.norm3 = _norm(norm_fn, planes)          # registered twice on purpose: the reference's
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)  # keys

    def forward(self, x):
        y = self.non_linearity(self.norm1(self.conv1(x)))
        y = self.non_linearity(self.norm2(self.conv2(y)))
        if self.downsample is not None:
            x = self.downsample(x)
        return self.non_linearity(x + y)


class BasicEncoder(nn.Module):
    """extractor.py:142-215: 7x7/2 stem, three residual stages (64, 96/2, 128/2), 1x1 head."""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0):
        super().__init__()
        self.norm_fn = norm_fn
        self.norm1 = _norm(norm_fn, 64, groups=8)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(ResidualBlock(64, 64, norm_fn, 1), ResidualBlock(64, 64, norm_fn, 1))
        self.layer2 = nn.Sequential(ResidualBlock(64, 96, norm_fn, 2), ResidualBlock(96, 96, norm_fn, 1))
        self.layer3 = nn.Sequential(ResidualBlock(96, 128, norm_fn, 2), ResidualBlock(128, 128, norm_fn, 1))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def _engine_ok(self, x) -> bool:
        """The native encoder (raft_encoder_engine.py): frozen parameters, eval mode, HIP float32, sides multiples of 8; a
        refused forward is reported once (`_lib.engine_gate`)."""
        extra = None if self.norm_fn in ("instance", "batch") else f"norm_fn {self.norm_fn!r} is not instance / batch"
        return L.engine_gate(self, x, 8, extra=extra if os.environ.get("UFR_ENGINE", "1") == "1" and x.is_cuda else None)

    def forward(self, x, stacked=None):
        """`stacked`: the pair already concatenated along the batch (RAFT.forward's normalised stack), instead of a torch.cat here."""
        pair = isinstance(x, (tuple, list))
        if pair:
            n = x[0].shape[0]
            x = stacked if stacked is not None else torch.cat(x, dim=0)
        if self._engine_ok(x):
            from ..raft_encoder_engine import encode      # stem, residual stages and head on the native engine
            x = encode(self, x)
            return torch.split(x, [n, n], dim=0) if pair else x
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.conv2(self.layer3(self.layer2(self.layer1(x))))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        return torch.split(x, [n, n], dim=0) if pair else x


class _ConvexUpsample(torch.autograd.Function):
    """RAFT.upsample_flow as one kernel forward, two backward (csrc/convex_upsample.hip)."""

    @staticmethod
    def forward(ctx, flow, mask):
        N, _, H, W = flow.shape
        if mask.shape != (N, 576, H, W):
            raise RuntimeError("upsample_flow: mask must be [N, 9*8*8, H, W]")
        up = torch.empty(N, 2, 8 * H, 8 * W, dtype=flow.dtype, device=flow.device)
        L.check(L.lib().ufr_convex_upsample_forward(L.ptr(flow), L.ptr(mask), L.ptr(up), N, H, W, L.stream()),
                "convex upsample forward")
        ctx.save_for_backward(flow, mask)
        return up

    @staticmethod
    def backward(ctx, g_up):
        flow, mask = ctx.saved_tensors
        N, _, H, W = flow.shape
        g_flow, g_mask = torch.empty_like(flow), torch.empty_like(mask)
        ws = torch.empty(N, 2, 9, H, W, dtype=flow.dtype, device=flow.device)
        L.check(L.lib().ufr_convex_upsample_backward(L.ptr(flow), L.ptr(mask), L.ptr(g_up.contiguous()), L.ptr(g_flow),
                                                     L.ptr(g_mask), L.ptr(ws), N, H, W, L.stream()), "convex upsample backward")
        return g_flow, g_mask


class FlowHead(nn.Module):
    """update.py:6-14."""

    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv2d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = _FlowConv(hidden_dim, 2, 3, padding=1)      # 2 output channels: single-pass kernel (small_cout.hip)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.conv2(conv_relu(x, self.conv1))


class _GruGates(torch.autograd.Function):
    """z = sigmoid(zr[:, :Ch]); rh = sigmoid(zr[:, Ch:]) * h   (csrc/gru.hip)."""

    @staticmethod
    def forward(ctx, zr_pre, h):
        zr_pre, h = zr_pre.contiguous(), h.contiguous()
        B, Ch, H, W = h.shape
        z, rh = torch.empty_like(h), torch.empty_like(h)
        L.check(L.lib().ufr_gru_gates_forward(L.ptr(zr_pre), L.ptr(h), L.ptr(z), L.ptr(rh), B, Ch, H * W,
                                              Ch * H * W, L.stream()), "gru gates forward")
        ctx.save_for_backward(zr_pre, h)
        return z, rh

    @staticmethod
    def backward(ctx, g_z, g_rh):
        zr_pre, h = ctx.saved_tensors
        B, Ch, H, W = h.shape
        g_z = torch.zeros_like(h) if g_z is None else g_z.contiguous()
        g_rh = torch.zeros_like(h) if g_rh is None else g_rh.contiguous()
        g_zr, g_h = torch.empty_like(zr_pre), torch.empty_like(h)
        L.check(L.lib().ufr_gru_gates_backward(L.ptr(zr_pre), L.ptr(h), L.ptr(g_z), L.ptr(g_rh), L.ptr(g_zr),
                                               L.ptr(g_h), B, Ch, H * W, Ch * H * W, L.stream()), "gru gates backward")
        return g_zr, g_h


class _GruBlend(torch.autograd.Function):
    """h' = (1 - z)*h + z*tanh(q_pre)   (csrc/gru.hip)."""

    @staticmethod
    def forward(ctx, q_pre, z, h):
        q_pre, z, h = q_pre.contiguous(), z.contiguous(), h.contiguous()
        out = torch.empty_like(h)
        L.check(L.lib().ufr_gru_blend_forward(L.ptr(q_pre), L.ptr(z), L.ptr(h), L.ptr(out), h.numel(), L.stream()),
                "gru blend forward")
        ctx.save_for_backward(q_pre, z, h)
        return out

    @staticmethod
    def backward(ctx, g):
        q_pre, z, h = ctx.saved_tensors
        g = g.contiguous()
        g_q, g_z, g_h = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        L.check(L.lib().ufr_gru_blend_backward(L.ptr(q_pre), L.ptr(z), L.ptr(h), L.ptr(g), L.ptr(g_q), L.ptr(g_z),
                                               L.ptr(g_h), h.numel(), L.stream()), "gru blend backward")
        return g_q, g_z, g_h


class SepConvGRU(nn.Module):
    """update.py:35-73: a horizontal (1x5) then a vertical (5x1) convolutional GRU step.
    On the device the z and r convolutions of a half-step run as ONE convolution (weights stacked
    once per forward, `cache`) and the gate arithmetic is two fused kernels instead of ten."""

    def __init__(self, hidden_dim=128, input_dim=192 + 128):
        super().__init__()
        self._pads = {"1": (0, 2), "2": (2, 0)}
        for tag, k, pad in (("1", (1, 5), (0, 2)), ("2", (5, 1), (2, 0))):
            for gate in "zrq":
                setattr(self, f"conv{gate}{tag}", nn.Conv2d(hidden_dim + input_dim, hidden_dim, k, padding=pad))

    def _half(self, h, x, tag, cache):
        convz, convr, convq = (getattr(self, f"conv{g}{tag}") for g in "zrq")
        if not (h.is_cuda and h.dtype == torch.float32):
            raise RuntimeError("SepConvGRU runs on float32 HIP tensors (no CPU path in this build)")
        if cache is None or tag not in cache:
            fused = (torch.cat([convz.weight, convr.weight]), torch.cat([convz.bias, convr.bias]))
            if cache is not None:
                cache[tag] = fused
        else:
            fused = cache[tag]
        zr = F.conv2d(torch.cat([h, x], dim=1), fused[0], fused[1], padding=self._pads[tag])
        z, rh = _GruGates.apply(zr, h)
        return _GruBlend.apply(convq(torch.cat([rh, x], dim=1)), z, h)

    def forward(self, h, x, cache=None):
        return self._half(self._half(h, x, "1", cache), x, "2", cache)


class BasicMotionEncoder(nn.Module):
    """update.py:94-120."""

    def __init__(self, args):
        super().__init__()
        cor_planes = args.corr_levels * (2 * args.corr_radius + 1) ** 2
        self.args = args
        self.convc1 = nn.Conv2d(cor_planes, 256, 1, padding=0)
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.conv = nn.Conv2d(64 + 192, 128 - 2, 3, padding=1)

    def forward(self, flow, corr):
        cor = conv_relu(corr, self.convc1)
        if not getattr(self.args, "update_no_motion_downsampling", False):
            cor = conv_relu(cor, self.convc2)
        flo = conv_relu(conv_relu(flow, self.convf1), self.convf2)
        out = conv_relu(torch.cat([cor, flo], dim=1), self.conv)
        return torch.cat([out, flow], dim=1)


class BasicUpdateBlock(nn.Module):
    """update.py:139-162."""

    def __init__(self, args, hidden_dim=128):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        self.gru = SepConvGRU(hidden_dim=hidden_dim, input_dim=128 + hidden_dim)
        self.flow_head = FlowHead(hidden_dim, hidden_dim=256)
        self.mask = nn.Sequential(nn.Conv2d(128, 256, 3, padding=1), nn.ReLU(inplace=True),
                                  nn.Conv2d(256, 64 * 9, 1, padding=0))

    def forward(self, net, inp, corr, flow, want_mask=True, cache=None):
        motion_features = self.encoder(flow, corr)
        net = self.gru(net, torch.cat([inp, motion_features], dim=1), cache)
        delta_flow = self.flow_head(net)
        # the x8 convex-upsampling mask only matters for the iteration whose flow is returned
        mask = 0.25 * self.mask(net) if want_mask else None
        return net, mask, delta_flow


_CONTEXT_STREAMS: dict = {}


def _context_stream(device):
    key = torch.device(device)
    if key not in _CONTEXT_STREAMS:
        _CONTEXT_STREAMS[key] = torch.cuda.Stream(device=key)
    return _CONTEXT_STREAMS[key]


def coords_grid(batch, ht, wd, device):
    """utils/utils.py:80-83: channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack((xs, ys), dim=0).float()[None].repeat(batch, 1, 1, 1)


class RAFT(nn.Module):
    def __init__(self, args, return_feat_maps: bool = False):
        super().__init__()
        if return_feat_maps:
            raise NotImplementedError("feature-map capture is analysis-only (out of scope)")
        if getattr(args, "small", False) or getattr(args, "flowNetCEnc", False) or getattr(args, "no_separate_context", False):
            raise NotImplementedError("only the raft-things configuration (BasicEncoder x2) is in scope")
        self.args = args
        self.hidden_dim = self.context_dim = 128
        args.corr_radius = 4                                           # raft.py:43-52 inject defaults
        for k, v in (("dropout", 0), ("alternate_corr", False), ("compute_spatial", False),
                     ("corr_levels", 4), ("iters", 12), ("fnorm", "instance"), ("cnorm", "batch")):
            if not hasattr(args, k):
                setattr(args, k, v)
        self.fnet = BasicEncoder(output_dim=256, norm_fn=args.fnorm, dropout=args.dropout)
        self.cnet = BasicEncoder(output_dim=256, norm_fn=args.cnorm, dropout=args.dropout)
        self.update_block = BasicUpdateBlock(args, hidden_dim=128)

    def products(self) -> int:
        """bf16 products per float32 product of the engines' convolutions: 6 unless the caller opted into reduced precision with
        BOTH `args.mixed_precision` (the reference's flag, models/utils_model.py:51) and UFR_RAFT_PRECISION=bf16 | bf16x3."""
        want = os.environ.get("UFR_RAFT_PRECISION", "fp32").lower()
        if want not in ("fp32", "float32", "bf16", "bf16x3"):
            raise ValueError(f"UFR_RAFT_PRECISION={want!r}: fp32, bf16 or bf16x3")
        if not getattr(self.args, "mixed_precision", False) or want in ("fp32", "float32"):
            return 6
        return 1 if want == "bf16" else 3

    def _engine_ok(self, net, H, W) -> bool:
        """The native refinement loop (raft_engine.py) serves the attack's configuration: frozen parameters, eval mode, HIP
        float32 tensors, the raft-things update block, frame sides that are multiples of 8 (UFR_ENGINE=0 switches it off);
        a refused forward is reported once (`_lib.engine_gate`)."""
        extra = None
        if getattr(self.args, "update_no_motion_downsampling", False) or self.args.corr_levels != 4 or self.args.corr_radius != 4:
            extra = "not the raft-things update block (corr_levels 4, corr_radius 4, motion downsampling)"
        return L.engine_gate(self, net, 8, 8, extra=extra if os.environ.get("UFR_ENGINE", "1") == "1" and net.is_cuda else None)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    @staticmethod
    def upsample_flow(flow, mask):
        """raft.py:111-122: [N,2,H,W] -> [N,2,8H,8W], softmax-weighted combination of the 3x3 coarse
        neighbours.  One fused kernel on the device (csrc/convex_upsample.hip); the torch spelling elsewhere."""
        if flow.is_cuda and flow.dtype == torch.float32 and mask.dtype == torch.float32:
            return _ConvexUpsample.apply(flow.contiguous(), mask.contiguous())
        N, _, H, W = flow.shape
        mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
        up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
        up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
        return up.reshape(N, 2, 8 * H, 8 * W)

    def forward(self, image1, image2, iters=12, flow_init=None, test_mode=False):
        iters = self.args.iters                                        # raft.py:126 (argument ignored)
        from ..raft_glue import normalize_pair
        stack = normalize_pair(image1, image2)                         # both frames in one pass, as the stack the feature encoder takes
        if stack is not None:
            image1, image2 = stack[:image1.shape[0]], stack[image1.shape[0]:]
        else:
            image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
            image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
        from .. import igemm as _ig
        prec = _ig.products(self.products())      # the engines built / looked up inside take this many products (6 = float32-accurate)

        def context():
            with prec:
                net, inp = torch.split(self.cnet(image1), [128, 128], dim=1)
            return torch.tanh(net), torch.relu(inp)

        # The context encoder reads frame 1 only and meets the feature encoder's results in the update block (raft.py:176-184).  On
        # the native path it runs on a second HIP stream: its normalisation kernels are HBM-bound, the feature encoder's
        # convolutions MFMA-bound, and the adjoints (autograd replays a node on the stream it was recorded on) run side by side
        # as well.  UFR_RAFT_STREAMS=0: one stream.
        fork = (image1.is_cuda and L.engine_refusal(self, image1, 8) is None and os.environ.get("UFR_ENGINE", "1") == "1"
                and os.environ.get("UFR_RAFT_STREAMS", "1") != "0")
        if fork:
            main, side = torch.cuda.current_stream(image1.device), _context_stream(image1.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                net, inp = context()
        with prec:
            fmap1, fmap2 = self.fnet([image1, image2], stacked=stack)
        fmap1, fmap2 = fmap1.float().contiguous(), fmap2.float().contiguous()
        if self.args.alternate_corr:
            corr_fn = AlternateCorrBlock(fmap1, fmap2, radius=self.args.corr_radius, share_grad=True)
        else:
            corr_fn = CorrBlock(fmap1, fmap2, num_levels=self.args.corr_levels, radius=self.args.corr_radius)
        if fork:
            main.wait_stream(side)
            net.record_stream(main), inp.record_stream(main)
        else:
            net, inp = context()

        N, _, H, W = image1.shape
        if test_mode and flow_init is None and self._engine_ok(net, H, W):
            # the 12 iterations as one explicit schedule on the native engine (raft_engine.py): same operands, same result
            from ..raft_engine import refine
            with prec:
                flow_lr, up_mask = refine(self, net.contiguous(), inp.contiguous(), corr_fn, H, W)
            return flow_lr, self.upsample_flow(flow_lr, up_mask)
        coords0 = coords_grid(N, H // 8, W // 8, image1.device)
        coords1 = coords0.clone()
        if flow_init is not None:
            coords1 = coords1 + flow_init
        flow_predictions, flow_up = [], None
        gru_cache = {}                      # stacked z|r gate weights, built once per forward
        for it in range(iters):
            coords1 = coords1.detach()
            corr = corr_fn(coords1)
            want = (not test_mode) or it == iters - 1
            net, up_mask, delta_flow = self.update_block(net, inp, corr, coords1 - coords0, want_mask=want,
                                                         cache=gru_cache)
            coords1 = coords1 + delta_flow
            if want:
                flow_up = self.upsample_flow(coords1 - coords0, up_mask)
                flow_predictions.append(flow_up)
        if test_mode:
            return coords1 - coords0, flow_up
        return flow_predictions

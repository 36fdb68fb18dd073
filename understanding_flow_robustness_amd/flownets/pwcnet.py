"""PWC-DC-Net (9.4 M parameters) on the gfx950 correlation kernel.

Behavioural mirror of models/PWCNet.py:52-367 (layer names kept, so the reference's
`pwc_net_chairs.pth.tar` / adversarially trained state_dicts load unchanged): 6-level siamese
pyramid, per level  warp(second-frame features, upsampled flow * scale) -> 9x9 correlation (`/C`)
-> LeakyReLU -> DenseNet decoder -> flow, then the dilated context network; output 20 * upsample x4.
The two pyramids run as one batch of 2B images.
"""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _lib as L
from ..band_conv import FlowHead, FlowUpsample, conv_leaky
from ..cone import ConeSpec
from .flownetc import correlate as _correlate


def _conv(cin, cout, k=3, stride=1, padding=1, dilation=1):
    """PWCNet.py:17-30."""
    return nn.Sequential(nn.Conv2d(int(cin), int(cout), k, stride, padding, dilation, bias=True), nn.LeakyReLU(0.1))


def correlate(input1, input2):
    """PWCNet.py:42-50: 9x9 displacements, dilation_patch 1, divided by C."""
    return _correlate(input1, input2, patch_size=9, dilation_patch=1)


_WARP_WORKSPACES = L.LruDict(32)       # (device, stream, B, H, W) -> the owner-computes adjoint's table of sampling boxes


def warp_backward(x, flo, g, gx, gf):
    """The warp's adjoint (csrc/pwc_warp.hip): owner-computes, no float atomics, grad_x written once -- `gx` need not be zeroed.
    UFR_PWC_WARP_OWNER=0 keeps the scatter with float atomics (the A/B switch of tests/test_pwc_warp_gpu.py).  The workspace is written
    and read by one call in stream order: one per (stream, shape)."""
    B, Cn, H, W = x.shape
    lib = L.lib()
    if os.environ.get("UFR_PWC_WARP_OWNER", "1") == "0":
        L.check(lib.ufr_pwc_warp_backward(L.ptr(x), L.ptr(flo), L.ptr(g), L.ptr(gx), L.ptr(gf), B, Cn, H, W, L.stream()), "pwc warp backward")
        return
    key = (x.device, torch.cuda.current_stream(x.device).cuda_stream, B, H, W)
    ws = _WARP_WORKSPACES.get(key)
    if ws is None:
        nbytes = int(lib.ufr_pwc_warp_backward_workspace_bytes(B, H, W))
        ws = _WARP_WORKSPACES[key] = (torch.empty((nbytes + 15) // 16 * 4, dtype=torch.int32, device=x.device), nbytes)
    L.check(lib.ufr_pwc_warp_backward_owner(L.ptr(x), L.ptr(flo), L.ptr(g), L.ptr(gx), L.ptr(gf), L.ptr(ws[0]), ws[1], B, Cn, H, W,
                                            L.stream()), "pwc warp backward")


class _PwcWarp(torch.autograd.Function):
    """`warp` as one kernel forward, one backward (csrc/pwc_warp.hip)."""

    @staticmethod
    def forward(ctx, x, flo):
        B, Cn, H, W = x.shape
        out = torch.empty_like(x)
        L.check(L.lib().ufr_pwc_warp_forward(L.ptr(x), L.ptr(flo), L.ptr(out), B, Cn, H, W, L.stream()), "pwc warp forward")
        ctx.save_for_backward(x, flo)
        return out

    @staticmethod
    def backward(ctx, g):
        x, flo = ctx.saved_tensors
        B, Cn, H, W = x.shape
        gx, gf = torch.empty_like(x), torch.empty_like(flo)
        with torch.cuda.device(x.device):
            warp_backward(x, flo, g.contiguous(), gx, gf)
        return gx, gf


def warp(x, flo):
    """PWCNet.py:164-204: bilinear backward warp (grid_sample, default align_corners=False, grid
    normalised with (W-1)) times the validity mask `warp(ones) >= 0.0001`.  Fused kernel on the device."""
    if x.is_cuda and x.dtype == torch.float32 and flo.dtype == torch.float32:
        return _PwcWarp.apply(x.contiguous(), flo.contiguous())
    return _warp_torch(x, flo)


def _warp_torch(x, flo):
    """The reference's spelling (also the checker of tests/test_pwc_warp_gpu.py)."""
    B, _, H, W = x.shape
    xx = torch.arange(W, device=x.device, dtype=x.dtype).view(1, 1, 1, W)
    yy = torch.arange(H, device=x.device, dtype=x.dtype).view(1, 1, H, 1)
    vx = 2.0 * (xx + flo[:, 0:1]) / max(W - 1, 1) - 1.0
    vy = 2.0 * (yy + flo[:, 1:2]) / max(H - 1, 1) - 1.0
    vgrid = torch.cat((vx, vy), 1).permute(0, 2, 3, 1)
    output = F.grid_sample(x, vgrid, align_corners=False)
    mask = F.grid_sample(torch.ones_like(x[:, :1]), vgrid, align_corners=False)
    return output * (mask >= 0.0001).to(x.dtype)


class PWCDCNet(nn.Module):
    _PYRAMID = ((3, 16, "1a", "1aa", "1b"), (16, 32, "2a", "2aa", "2b"), (32, 64, "3a", "3aa", "3b"),
                (64, 96, "4a", "4aa", "4b"), (96, 128, "5a", "5aa", "5b"), (128, 196, "6aa", "6a", "6b"))
    _FLOW_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}     # PWCNet.py:286,301,316,332

    def __init__(self, md=4, pretrained=False, return_feat_maps=False):
        super().__init__()
        if return_feat_maps:
            raise NotImplementedError("feature-map capture is analysis-only (out of scope)")
        for cin, cout, first, second, third in self._PYRAMID:
            setattr(self, "conv" + first, _conv(cin, cout, 3, 2))
            setattr(self, "conv" + second, _conv(cout, cout, 3, 1))
            setattr(self, "conv" + third, _conv(cout, cout, 3, 1))
        nd = (2 * md + 1) ** 2
        dd = np.cumsum([128, 128, 96, 64, 32])
        feat = {6: 0, 5: 128, 4: 96, 3: 64, 2: 32}
        for lvl in (6, 5, 4, 3, 2):
            od = nd if lvl == 6 else nd + feat[lvl] + 4
            for i, (extra, cout) in enumerate(zip((0, dd[0], dd[1], dd[2], dd[3]), (128, 128, 96, 64, 32))):
                setattr(self, f"conv{lvl}_{i}", _conv(od + extra, cout))
            setattr(self, f"predict_flow{lvl}", FlowHead(int(od + dd[4]), 2, 3, 1, 1, bias=True))
            setattr(self, f"deconv{lvl}", FlowUpsample(2, 2, 4, 2, 1, bias=True))
            if lvl > 2:
                setattr(self, f"upfeat{lvl}", nn.ConvTranspose2d(int(od + dd[4]), 2, 4, 2, 1, bias=True))
        od = nd + 32 + 4
        for i, (cin, cout, dil) in enumerate(((od + dd[4], 128, 1), (128, 128, 2), (128, 128, 4), (128, 96, 8),
                                              (96, 64, 16), (64, 32, 1)), start=1):
            setattr(self, f"dc_conv{i}", _conv(cin, cout, 3, 1, dil, dil))
        self.dc_conv7 = FlowHead(32, 2, 3, 1, 1, bias=True)
        for m in self.modules():                                   # PWCNet.py:154-158
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.kaiming_normal_(m.weight.data, mode="fan_in")
                if m.bias is not None:
                    m.bias.data.zero_()

    def _cl(self, name, x):
        """One conv + bias + LeakyReLU block; bias and activation are one in-place pass on the device."""
        return conv_leaky(x, getattr(self, name))

    def _decode(self, lvl, x):
        for i in range(5):                                         # DenseNet connections
            x = torch.cat((self._cl(f"conv{lvl}_{i}", x), x), 1)
        return x, getattr(self, f"predict_flow{lvl}")(x)

    # pyramid levels 1-2 are a purely convolutional prefix at 1/2 and 1/4 resolution (16 / 32 channels: few FLOPs,
    # 6 GB of activation traffic per iteration at batch 8): the patch attack runs them on a window (cone.py).
    # One tap: the level-2 features of both frames; level 1 feeds nothing else.
    CONE = ConeSpec(layers=((3, 2, 1), (3, 1, 1), (3, 1, 1), (3, 2, 1), (3, 1, 1), (3, 1, 1)), taps=(5,), frames=(2,))

    def encode(self, x):
        """Levels 1-2 of the feature pyramid on a stack of raw frames [N,3,h,w] -> [level-2 features]."""
        native = self._native_prefix(x)
        if native is not None:
            return [native]
        x = x.flip(1)                                              # RGB -> BGR (PWCNet.py:230-231)
        for _, _, first, second, third in self._PYRAMID[:2]:
            x = self._cl("conv" + third, self._cl("conv" + second, self._cl("conv" + first, x)))
        return [x]

    def _native_prefix(self, x):
        """Levels 1-2 on the hand-written kernels (plane_graph.py: six 3x3 convolutions, forward and data gradient on the igemm,
        the BGR flip folded into conv1a's weights) whenever the network is frozen and in eval mode: the clean forward behind the
        attack's target, the validation loop, the full-frame iteration.  None = not served (training, other sizes, UFR_ENGINE=0)."""
        from ..plane_graph import PlaneGraph, graph_for, native_ok, run
        if not native_ok(self, x):
            return None
        n, _, H, W = x.shape

        def build():
            g = PlaneGraph(n, x.device)
            g.buffer("in0", H, W, 1)
            g.input("in0", 3)
            prev, level = "in0", 0
            for lvl, (_, cout, first, second, third) in enumerate(self._PYRAMID[:2], start=1):
                for j, name in enumerate((first, second, third)):
                    conv = getattr(self, "conv" + name)[0]
                    w = conv.weight.detach()
                    if lvl == 1 and j == 0:
                        w = w.flip(1)                               # x.flip(1) of the reference (PWCNet.py:230-231)
                    buf = f"l{lvl}_{j}"
                    g.buffer(buf, H >> lvl, W >> lvl, 1)
                    g.conv(w, conv.bias, (prev, 0, 1), (buf, 0), stride=2 if j == 0 else 1)
                    prev = buf
            g.tensor_output(prev, 32)
            return g.build()

        return run(graph_for(self, ("pyramid12", n, H, W, str(x.device)), build), x)[0]

    ENGINE = "pwc"                                                  # patch_attack.py: which native head serves this network

    def engine_available(self, H, W, device) -> bool:
        """Can the attack step keep this network's cached level-2 features inside the native head (pwc_engine.py)?"""
        import os
        frozen = not any(p.requires_grad for p in self.parameters())
        return (os.environ.get("UFR_ENGINE", "1") == "1" and not self.training and frozen
                and torch.device(device).type == "cuda" and H % 64 == 0 and W % 64 == 0)

    def _engine_ok(self, f2a) -> bool:
        """The native head (pwc_engine.py) serves the attack's configuration: frozen parameters, eval mode, HIP float32
        features, frame sides that are multiples of 64 (UFR_ENGINE=0 switches it off); a refused forward is reported once."""
        from .. import _lib as L
        return L.engine_gate(self, f2a, 64, 4)

    def head(self, f2a, f2b):
        """Pyramid levels 3-6, the coarse-to-fine decoder and the context network."""
        if self._engine_ok(f2a):
            from ..pwc_engine import engine_head
            flow2 = engine_head(self, f2a, f2b)
            return 20 * F.interpolate(flow2, scale_factor=4, mode="bilinear", align_corners=False)
        B = f2a.shape[0]
        x = torch.cat((f2a, f2b), 0)
        feats = [None, x]
        for _, _, first, second, third in self._PYRAMID[2:]:
            x = self._cl("conv" + third, self._cl("conv" + second, self._cl("conv" + first, x)))
            feats.append(x)
        return self._decoder(feats, B)

    def forward(self, im1, im2):
        B = im1.shape[0]
        (f2,) = self.encode(torch.cat((im1, im2), 0))
        return self.head(f2[:B], f2[B:])

    def _decoder(self, feats, B):
        c1 = {lvl: feats[lvl - 1][:B] for lvl in range(2, 7)}
        c2 = {lvl: feats[lvl - 1][B:] for lvl in range(2, 7)}

        corr = F.leaky_relu(correlate(c1[6].contiguous(), c2[6].contiguous()), 0.1)
        x, flow = self._decode(6, corr)
        flows = {6: flow}
        for lvl in (5, 4, 3, 2):
            up_flow = getattr(self, f"deconv{lvl + 1}")(flow)
            up_feat = getattr(self, f"upfeat{lvl + 1}")(x)
            warped = warp(c2[lvl], up_flow * self._FLOW_SCALE[lvl])
            corr = F.leaky_relu(correlate(c1[lvl].contiguous(), warped.contiguous()), 0.1)
            x, flow = self._decode(lvl, torch.cat((corr, c1[lvl], up_flow, up_feat), 1))
            flows[lvl] = flow
        x = self._cl("dc_conv4", self._cl("dc_conv3", self._cl("dc_conv2", self._cl("dc_conv1", x))))
        flow2 = flows[2] + self.dc_conv7(self._cl("dc_conv6", self._cl("dc_conv5", x)))
        up = lambda f: F.interpolate(f, scale_factor=4, mode="bilinear", align_corners=False)
        if self.training:
            return tuple(up(f) for f in (flow2, flows[3], flows[4], flows[5], flows[6]))
        return 20 * up(flow2)


def pwc_dc_net(path=None, return_feat_maps=False):
    """PWCNet.py:381-390."""
    model = PWCDCNet(return_feat_maps=return_feat_maps)
    if path is not None:
        data = torch.load(path, map_location="cpu")
        model.load_state_dict(data["state_dict"] if "state_dict" in data else data)
    return model

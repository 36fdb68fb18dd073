"""FlowNetC (39.2 M parameters) on the gfx950 correlation kernel.

Behavioural mirror of models/FlowNetC.py:11-197 + models/submodules.py:18-138 of the reference
(same layer names -> the reference's `FlowNet2-C_checkpoint.pth.tar` state_dict loads unchanged):
siamese conv1-3, 21x21 stride-2 correlation (`correlate`: view + /C), LeakyReLU(0.1), conv_redir,
conv3_1..conv6_1, coarse-to-fine refinement, flow2*20 upsampled x4.
The two siamese towers run as ONE batch of 2B images (same arithmetic per image, half the launches).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..band_conv import FlowHead, FlowUpsample, conv_leaky, flow_head, flow_upsample, window_correlation
from ..cone import ConeSpec
from ..spatial_correlation_sampler import spatial_correlation_sample

# models/FlowNetC.py:73-79 -- RGB mean, subtracted in float64 and cast back
_RGB_MEAN = (0.40066648, 0.39482617, 0.3784785)


class ConvLeaky(nn.Sequential):
    """Sequential(convolution with bias, LeakyReLU) -- same parameter names as the reference's blocks -- whose
    forward fuses bias + activation on the device (band_conv.conv_leaky)."""

    def forward(self, x):
        return conv_leaky(x, self)


def _conv(cin, cout, k=3, stride=1):
    """submodules.py:18-46 (batchNorm=False branch): conv + LeakyReLU(0.1)."""
    return ConvLeaky(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=True), nn.LeakyReLU(0.1, inplace=True))


def _deconv(cin, cout):
    """submodules.py:75-82."""
    return ConvLeaky(nn.ConvTranspose2d(cin, cout, 4, 2, 1, bias=True), nn.LeakyReLU(0.1, inplace=True))


def correlate(input1, input2, patch_size=21, dilation_patch=2, band=None, in_stride=8):
    """submodules.py:124-138: cost volume as a 4-D tensor [B, P*P, H, W], divided by C.  With a band that
    carries the prefix window, the adjoint is computed on the window's cells only."""
    if band is not None and band.cone_win is not None and input1.requires_grad:
        out = window_correlation(input1, input2, patch_size, dilation_patch, band, in_stride)
    else:
        out = spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=patch_size, stride=1,
                                         padding=0, dilation_patch=dilation_patch)
    b, ph, pw, h, w = out.size()
    return out.view(b, ph * pw, h, w) / input1.size(1)


class FlowNetC(nn.Module):
    # (name, in, out, kernel, stride) in the reference's construction order
    _ENCODER = (("conv1", 3, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2),
                ("conv_redir", 256, 32, 1, 1), ("conv3_1", 473, 256, 3, 1), ("conv4", 256, 512, 3, 2),
                ("conv4_1", 512, 512, 3, 1), ("conv5", 512, 512, 3, 2), ("conv5_1", 512, 512, 3, 1),
                ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1))
    _DECODER = (("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128), ("deconv2", 386, 64))
    _HEADS = (("predict_flow6", 1024), ("predict_flow5", 1026), ("predict_flow4", 770),
              ("predict_flow3", 386), ("predict_flow2", 194))
    _UPS = ("upsampled_flow6_to_5", "upsampled_flow5_to_4", "upsampled_flow4_to_3", "upsampled_flow3_to_2")

    def __init__(self, batchNorm=False, div_flow=20, return_feat_maps=False):
        super().__init__()
        if batchNorm:
            raise NotImplementedError("the attack path uses the batchNorm=False checkpoints")
        self.div_flow, self.return_feat_maps = div_flow, return_feat_maps
        for name, cin, cout, k, s in self._ENCODER:
            setattr(self, name, _conv(cin, cout, k, s))
        for name, cin, cout in self._DECODER:
            setattr(self, name, _deconv(cin, cout))
        for name, cin in self._HEADS:
            setattr(self, name, FlowHead(cin, 2, 3, 1, 1, bias=True))
        for name in self._UPS:
            setattr(self, name, FlowUpsample(2, 2, 4, 2, 1, bias=True))
        for m in self.modules():                       # FlowNetC.py:53-63
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                nn.init.uniform_(m.bias)
                nn.init.xavier_uniform_(m.weight)
        self.register_buffer("_mean64", torch.tensor(_RGB_MEAN, dtype=torch.float64).view(1, 3, 1, 1),
                             persistent=False)

    def normalize_correctly(self, im):
        """FlowNetC.py:73-79,93-94: float64 mean subtraction, then back to float32."""
        return (im.double() - self._mean64).float()

    # conv1-3 are a purely convolutional prefix: the patch attack runs them on a window around the patch
    # (cone.py, patch_attack.py); taps = conv2 (skip connection, first frame) and conv3 (both frames).
    CONE = ConeSpec(layers=((7, 2, 3), (5, 2, 2), (5, 2, 2)), taps=(1, 2), frames=(1, 2))

    def encode(self, x):
        """Siamese prefix on a stack of raw frames [N,3,h,w] (h, w multiples of 8): (conv2, conv3)."""
        x = self.normalize_correctly(x)
        stem = self._native_stem(x)
        if stem is not None:
            return stem
        c1 = self._cl("conv1", x)
        c2 = self._cl("conv2", c1)
        return c2, self._cl("conv3", c2)

    def _native_stem(self, x):
        """conv1-3 on the hand-written kernels (plane_graph.stem_graph: forward and data gradient on the igemm) whenever the
        network is frozen and in eval mode -- the clean forward that makes the attack's target, the validation loop, the
        full-frame attack iteration -- so that no vendor convolution runs for FlowNetC at all; None = not served (training,
        other sizes, UFR_ENGINE=0)."""
        from ..plane_graph import graph_for, native_ok, run, stem_graph
        if not native_ok(self, x):
            return None
        n, _, h, w = x.shape
        g = graph_for(self, ("stem", n, h, w, str(x.device)), lambda: stem_graph(self, n, h, w, 3, x.device))
        c2, c3 = run(g, x)
        return c2, c3

    def head(self, c2a, c3a, c3b, band=None):
        """Everything after the prefix: correlation, conv_redir, conv3_1..6_1, refinement -> flow.
        `band` (band_conv.Band): c3a / c3b differ from constants only inside a window whose 21x21
        correlation reach lies within the band, so conv3_1 / conv4 / conv4_1 / conv5 compute their data
        gradient on the band's columns only."""
        return self._rest(c2a, c3a, c3b, None, band)

    # head convolutions whose data gradient is banded, with the pixel stride of their input
    BAND_LAYERS = (("conv3_1", 8), ("conv4", 8), ("conv4_1", 16), ("conv5", 16))
    # pixels beyond the prefix window that the band's exact zone must reach (worst layer: conv5's input):
    # 20 cells of correlation displacement (160) + conv3_1 (8) + conv4 (8 left / 16 right) + conv4_1 (16)
    # + one /16 cell of inexact rim next to an interior band edge (16)
    BAND_REACH = 160 + 8 + 16 + 16 + 16

    def _cl(self, name, x, band=None, in_stride=0):
        """One `conv` / `deconv` block (convolution + bias + LeakyReLU), fused epilogue on the device."""
        return conv_leaky(x, getattr(self, name), band, in_stride, name)

    # blocks whose forward is incremental from the second iteration of an attack() call on: their inputs
    # change only inside the band (the changed columns of conv4_1's output stay 16 pixels inside it)
    INCREMENTAL_LAYERS = ("conv3_1", "conv4", "conv4_1")

    def forward(self, x1, x2, overwrite_feat_maps=None):
        if overwrite_feat_maps is not None:
            raise NotImplementedError("feature-map overwriting belongs to the analysis scripts (out of scope)")
        B = x1.shape[0]
        x = self.normalize_correctly(torch.cat((x1, x2), 0))
        stem = None if self.return_feat_maps else self._native_stem(x)
        if stem is not None:
            c1, (c2, c3) = None, stem
        else:
            c1 = self._cl("conv1", x)
            c2 = self._cl("conv2", c1)
            c3 = self._cl("conv3", c2)
        c2a, c3a, c3b = c2[:B], c3[:B], c3[B:]
        feats = [c1[:B], c2a, c3a, c1[B:], c2[B:], c3b] if self.return_feat_maps else None
        return self._rest(c2a, c3a, c3b, feats)

    def engine_available(self, H, W, device) -> bool:
        """Can the attack step keep this network's cached features inside the native head (patch_attack.py's windowed
        iteration)?  Same conditions as `_engine_ok`, asked before any feature exists."""
        import os
        frozen = not any(p.requires_grad for p in self.parameters())
        return (os.environ.get("UFR_ENGINE", "1") == "1" and not self.training
                and not self.return_feat_maps and frozen and torch.device(device).type == "cuda" and H % 64 == 0 and W % 64 == 0)

    def _engine_ok(self, c2a, feats, band):
        """The native head (flownetc_engine.py) serves the attack's configuration: frozen parameters, eval mode, HIP
        float32 features, frame sides that are multiples of 64; a refused forward is reported once (`_lib.engine_gate`)."""
        if feats is not None:                  # return_feat_maps: the analysis scripts' torch spelling, by request
            return False
        from .. import _lib as L
        return L.engine_gate(self, c2a, 64, 4)

    def _rest(self, c2a, c3a, c3b, feats, band=None):
        if self._engine_ok(c2a, feats, band):
            from ..flownetc_engine import engine_head
            flow2 = engine_head(self, c2a, c3a, c3b, band)
            return F.interpolate(flow2 * self.div_flow, scale_factor=4, mode="bilinear", align_corners=False)
        out_corr = correlate(c3a.contiguous(), c3b.contiguous(), band=band)
        if feats is not None:
            feats.append(out_corr.clone())
        out_corr = F.leaky_relu(out_corr, 0.1)
        in_conv3_1 = torch.cat((self._cl("conv_redir", c3a), out_corr), 1)

        c3_1 = self._cl("conv3_1", in_conv3_1, band, 8)
        c4 = self._cl("conv4_1", self._cl("conv4", c3_1, band, 8), band, 16)
        c5 = self._cl("conv5_1", self._cl("conv5", c4, band, 16))
        c6 = self._cl("conv6_1", self._cl("conv6", c5))

        flow6 = flow_head(c6, self.predict_flow6)
        cat5 = torch.cat((c5, self._cl("deconv5", c6), flow_upsample(flow6, self.upsampled_flow6_to_5)), 1)
        flow5 = flow_head(cat5, self.predict_flow5)
        cat4 = torch.cat((c4, self._cl("deconv4", cat5), flow_upsample(flow5, self.upsampled_flow5_to_4)), 1)
        flow4 = flow_head(cat4, self.predict_flow4)
        cat3 = torch.cat((c3_1, self._cl("deconv3", cat4), flow_upsample(flow4, self.upsampled_flow4_to_3)), 1)
        flow3 = flow_head(cat3, self.predict_flow3)
        cat2 = torch.cat((c2a, self._cl("deconv2", cat3), flow_upsample(flow3, self.upsampled_flow3_to_2)), 1)
        flow2 = flow_head(cat2, self.predict_flow2)

        up = lambda f: F.interpolate(f * self.div_flow, scale_factor=4, mode="bilinear", align_corners=False)
        if self.training:
            return tuple(up(f) for f in (flow2, flow3, flow4, flow5, flow6))
        return (up(flow2), feats) if self.return_feat_maps else up(flow2)

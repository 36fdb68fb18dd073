"""FlowNet2 (162.5 M parameters): FlowNetC -> warp -> FlowNetS -> warp -> FlowNetS, FlowNetSD, fusion --
on the gfx950 correlation / Resample2d / ChannelNorm kernels.

Behavioural mirror of models/flownet2_models.py:14-205 and models/flownet2/{FlowNetC,FlowNetS,
FlowNetSD,FlowNetFusion}.py with the reference's layer names (`FlowNet2_checkpoint.pth.tar` loads
unchanged).  Quirks kept: FlowNetSD's flow is DIVIDED by div_flow (:176), FlowNetS-2 / FlowNetSD
flows are upsampled with `nearest`, FlowNetS's flow up-convolutions have no bias.
"""
from __future__ import annotations

import os

import torch
from .. import _lib as _L
from ..plane_graph import graph_for as _graph_for, native_ok as _native_ok, stem_graph as _prefix_graph
import torch.nn as nn
import torch.nn.functional as F

from ..band_conv import FlowHead, FlowUpsample
from ..channelnorm_package.channelnorm import ChannelNorm
from ..resample2d_package.resample2d import Resample2d
from .flownetc import _RGB_MEAN, _conv, _deconv, correlate


def _i_conv(cin, cout, k=3, stride=1, bias=True):
    """submodules.py:48-72 (batchNorm=False): convolution without activation."""
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=bias))


def _flow(cin):
    return FlowHead(cin, 2, 3, 1, 1, bias=True)


def _xavier(module):
    for m in module.modules():                              # every sub-net's __init__ tail
        if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
            if m.bias is not None:
                nn.init.uniform_(m.bias)
            nn.init.xavier_uniform_(m.weight)


class _Refinement(nn.Module):
    """The coarse-to-fine decoder shared by FlowNetC and FlowNetS (levels 6 -> 2)."""

    def _build_refinement(self, up_bias):
        for name, cin, cout in (("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128),
                                ("deconv2", 386, 64)):
            setattr(self, name, _deconv(cin, cout))
        for name, cin in (("predict_flow6", 1024), ("predict_flow5", 1026), ("predict_flow4", 770),
                          ("predict_flow3", 386), ("predict_flow2", 194)):
            setattr(self, name, _flow(cin))
        for name in ("upsampled_flow6_to_5", "upsampled_flow5_to_4", "upsampled_flow4_to_3", "upsampled_flow3_to_2"):
            setattr(self, name, FlowUpsample(2, 2, 4, 2, 1, bias=up_bias))

    def _engine_ok(self, c2a) -> bool:
        """Frozen parameters, eval mode, HIP float32 features, frame sides that are multiples of 64 (UFR_ENGINE=0 switches off);
        a refused forward is reported once (`_lib.engine_gate`)."""
        from .. import _lib as L
        return L.engine_gate(self, c2a, 64, 4)

    def _refine(self, c6, skips):
        """skips = (conv5, conv4, conv3, conv2) features; returns flow2."""
        flow = self.predict_flow6(c6)
        x = c6
        for lvl, skip in zip((5, 4, 3, 2), skips):
            up = getattr(self, f"upsampled_flow{lvl + 1}_to_{lvl}")(flow)
            x = torch.cat((skip, getattr(self, f"deconv{lvl}")(x), up), 1)
            flow = getattr(self, f"predict_flow{lvl}")(x)
        return flow


class FlowNetC(_Refinement):
    """models/flownet2/FlowNetC.py:10-131: 6-channel input, returns a tuple like the reference."""

    def __init__(self, batchNorm=False, div_flow=20):
        super().__init__()
        assert not batchNorm
        self.div_flow = div_flow
        for name, cin, cout, k, s in (("conv1", 3, 64, 7, 2), ("conv2", 64, 128, 5, 2), ("conv3", 128, 256, 5, 2),
                                      ("conv_redir", 256, 32, 1, 1), ("conv3_1", 473, 256, 3, 1),
                                      ("conv4", 256, 512, 3, 2), ("conv4_1", 512, 512, 3, 1),
                                      ("conv5", 512, 512, 3, 2), ("conv5_1", 512, 512, 3, 1),
                                      ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1)):
            setattr(self, name, _conv(cin, cout, k, s))
        self._build_refinement(up_bias=True)
        _xavier(self)

    def forward(self, x):
        B = x.shape[0]
        both = torch.cat((x[:, 0:3], x[:, 3:]), 0)
        if _native_ok(self, x):                # conv1-3 of both frames on the igemm too (plane_graph.py)
            from ..plane_graph import run
            g = _graph_for(self, ("prefix", 2 * B, x.shape[2], x.shape[3], str(x.device)),
                           lambda: _prefix_graph(self, 2 * B, x.shape[2], x.shape[3], 3, x.device))
            c2, c3 = run(g, both)
        else:
            c2 = self.conv2(self.conv1(both))
            c3 = self.conv3(c2)
        c2a, c3a, c3b = c2[:B], c3[:B], c3[B:]
        if self._engine_ok(c2a):
            # the same layers under the same names as flownets/flownetc.py: everything behind conv3 on the native head
            # (flownetc_engine.py: implicit-GEMM convolutions and the cost volume on the matrix cores, forward and data gradient)
            from ..flownetc_engine import engine_head
            return (engine_head(self, c2a.contiguous(), c3a.contiguous(), c3b.contiguous()),)
        corr = F.leaky_relu(correlate(c3a.contiguous(), c3b.contiguous()), 0.1)
        c3_1 = self.conv3_1(torch.cat((self.conv_redir(c3a), corr), 1))
        c4 = self.conv4_1(self.conv4(c3_1))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return (self._refine(c6, (c5, c4, c3_1, c2a)),)


class FlowNetS(_Refinement):
    """models/flownet2/FlowNetS.py:15-104."""

    def __init__(self, input_channels=12, batchNorm=False):
        super().__init__()
        assert not batchNorm
        for name, cin, cout, k, s in (("conv1", input_channels, 64, 7, 2), ("conv2", 64, 128, 5, 2),
                                      ("conv3", 128, 256, 5, 2), ("conv3_1", 256, 256, 3, 1),
                                      ("conv4", 256, 512, 3, 2), ("conv4_1", 512, 512, 3, 1),
                                      ("conv5", 512, 512, 3, 2), ("conv5_1", 512, 512, 3, 1),
                                      ("conv6", 512, 1024, 3, 2), ("conv6_1", 1024, 1024, 3, 1)):
            setattr(self, name, _conv(cin, cout, k, s))
        self._build_refinement(up_bias=False)
        _xavier(self)

    def forward(self, x):
        if _native_ok(self, x):                # conv1-3 on the igemm (plane_graph.py), everything behind on the native head
            from ..flownetc_engine import engine_head
            from ..plane_graph import run
            g = _graph_for(self, ("prefix", x.shape[0], x.shape[2], x.shape[3], str(x.device)),
                           lambda: _prefix_graph(self, x.shape[0], x.shape[2], x.shape[3], x.shape[1], x.device))
            c2, c3 = run(g, x)
            return (engine_head(self, c2.contiguous(), c3.contiguous(), None),)
        c2 = self.conv2(self.conv1(x))
        if self._engine_ok(c2):                # everything behind conv3 on the native head (trunk form: no correlation)
            from ..flownetc_engine import engine_head
            return (engine_head(self, c2.contiguous(), self.conv3(c2).contiguous(), None),)
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        return (self._refine(c6, (c5, c4, c3, c2)),)


class FlowNet2S(FlowNetS):
    """The registry's `FlowNetS` (models/__init__.py:2 binds it to models/FlowNet2S.py:15-108): two frames in,
    its own RGB mean subtracted in float64 (:62-68), the FlowNetS trunk on cat(x1, x2), and in eval mode
    `upsample1(flow2 * 20)` (bilinear x4, :105-108); in training the five raw flows (:102-103).  Same layer names as
    the reference, so `FlowNet2-S_checkpoint.pth.tar` loads unchanged."""

    _MEAN = (0.4114511, 0.43205959, 0.45015125)

    def __init__(self, input_channels=6, batchNorm=False, return_feat_maps=False):
        super().__init__(input_channels=input_channels, batchNorm=batchNorm)
        self.return_feat_maps = return_feat_maps
        self.register_buffer("_mean64", torch.tensor(self._MEAN, dtype=torch.float64).view(1, 3, 1, 1),
                             persistent=False)

    def _refine_all(self, c6, skips):
        flows = [self.predict_flow6(c6)]
        x = c6
        for lvl, skip in zip((5, 4, 3, 2), skips):
            up = getattr(self, f"upsampled_flow{lvl + 1}_to_{lvl}")(flows[-1])
            x = torch.cat((skip, getattr(self, f"deconv{lvl}")(x), up), 1)
            flows.append(getattr(self, f"predict_flow{lvl}")(x))
        return flows[::-1]                                        # flow2 ... flow6

    def forward(self, x1, x2):
        x1 = (x1.double() - self._mean64).float()
        x2 = (x2.double() - self._mean64).float()
        x = torch.cat((x1, x2), dim=1)
        if not self.training and _native_ok(self, x):          # eval + frozen: the FlowNetS trunk on the native kernels (stem + head)
            flow2 = FlowNetS.forward(self, x)[0]
            up = F.interpolate(flow2 * 20, scale_factor=4, mode="bilinear", align_corners=False)
            return (up, []) if self.return_feat_maps else up
        c2 = self.conv2(self.conv1(x))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flows = self._refine_all(c6, (c5, c4, c3, c2))
        if self.training:
            return tuple(flows)
        up = F.interpolate(flows[0] * 20, scale_factor=4, mode="bilinear", align_corners=False)
        return (up, []) if self.return_feat_maps else up


class FlowNetSD(nn.Module):
    """models/flownet2/FlowNetSD.py:12-126: small-displacement net with inter-convolutions."""

    def __init__(self, batchNorm=False):
        super().__init__()
        assert not batchNorm
        for name, cin, cout, s in (("conv0", 6, 64, 1), ("conv1", 64, 64, 2), ("conv1_1", 64, 128, 1),
                                   ("conv2", 128, 128, 2), ("conv2_1", 128, 128, 1), ("conv3", 128, 256, 2),
                                   ("conv3_1", 256, 256, 1), ("conv4", 256, 512, 2), ("conv4_1", 512, 512, 1),
                                   ("conv5", 512, 512, 2), ("conv5_1", 512, 512, 1), ("conv6", 512, 1024, 2),
                                   ("conv6_1", 1024, 1024, 1)):
            setattr(self, name, _conv(cin, cout, 3, s))
        for name, cin, cout in (("deconv5", 1024, 512), ("deconv4", 1026, 256), ("deconv3", 770, 128),
                                ("deconv2", 386, 64)):
            setattr(self, name, _deconv(cin, cout))
        for name, cin, cout in (("inter_conv5", 1026, 512), ("inter_conv4", 770, 256), ("inter_conv3", 386, 128),
                                ("inter_conv2", 194, 64)):
            setattr(self, name, _i_conv(cin, cout))
        for name, cin in (("predict_flow6", 1024), ("predict_flow5", 512), ("predict_flow4", 256),
                          ("predict_flow3", 128), ("predict_flow2", 64)):
            setattr(self, name, _flow(cin))
        for name in ("upsampled_flow6_to_5", "upsampled_flow5_to_4", "upsampled_flow4_to_3", "upsampled_flow3_to_2"):
            setattr(self, name, FlowUpsample(2, 2, 4, 2, 1))
        _xavier(self)

    def forward(self, x):
        if _native_ok(self, x):                # the whole sub-network as one schedule on the native kernels (plane_graph.py)
            from ..plane_graph import run
            g = _graph_for(self, (x.shape[0], x.shape[2], x.shape[3], str(x.device)),
                           lambda: _sd_graph(self, x.shape[0], x.shape[2], x.shape[3], x.device))
            return (run(g, x)[0],)
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        c3 = self.conv3_1(self.conv3(c2))
        c4 = self.conv4_1(self.conv4(c3))
        c5 = self.conv5_1(self.conv5(c4))
        c6 = self.conv6_1(self.conv6(c5))
        flow, feat = self.predict_flow6(c6), c6
        for lvl, skip in zip((5, 4, 3, 2), (c5, c4, c3, c2)):
            up = getattr(self, f"upsampled_flow{lvl + 1}_to_{lvl}")(flow)
            feat = torch.cat((skip, getattr(self, f"deconv{lvl}")(feat), up), 1)
            flow = getattr(self, f"predict_flow{lvl}")(getattr(self, f"inter_conv{lvl}")(feat))
        return (flow,)


class FlowNetFusion(nn.Module):
    """models/flownet2/FlowNetFusion.py:12-71: full-resolution fusion of the two flow candidates."""

    def __init__(self, batchNorm=False):
        super().__init__()
        assert not batchNorm
        for name, cin, cout, s in (("conv0", 11, 64, 1), ("conv1", 64, 64, 2), ("conv1_1", 64, 128, 1),
                                   ("conv2", 128, 128, 2), ("conv2_1", 128, 128, 1)):
            setattr(self, name, _conv(cin, cout, 3, s))
        self.deconv1, self.deconv0 = _deconv(128, 32), _deconv(162, 16)
        self.inter_conv1, self.inter_conv0 = _i_conv(162, 32), _i_conv(82, 16)
        self.predict_flow2, self.predict_flow1, self.predict_flow0 = _flow(128), _flow(32), _flow(16)
        self.upsampled_flow2_to_1 = FlowUpsample(2, 2, 4, 2, 1)
        self.upsampled_flow1_to_0 = FlowUpsample(2, 2, 4, 2, 1)
        _xavier(self)

    def forward(self, x):
        if _native_ok(self, x):
            from ..plane_graph import run
            g = _graph_for(self, (x.shape[0], x.shape[2], x.shape[3], str(x.device)),
                           lambda: _fusion_graph(self, x.shape[0], x.shape[2], x.shape[3], x.device))
            return run(g, x)[0]
        c0 = self.conv0(x)
        c1 = self.conv1_1(self.conv1(c0))
        c2 = self.conv2_1(self.conv2(c1))
        flow2 = self.predict_flow2(c2)
        cat1 = torch.cat((c1, self.deconv1(c2), self.upsampled_flow2_to_1(flow2)), 1)
        flow1 = self.predict_flow1(self.inter_conv1(cat1))
        cat0 = torch.cat((c0, self.deconv0(cat1), self.upsampled_flow1_to_0(flow1)), 1)
        return self.predict_flow0(self.inter_conv0(cat0))


def _sd_graph(net, B, H, W, dev):
    """FlowNetSD (models/flownet2/FlowNetSD.py:12-126): concatK = [convK_1 | deconvK | flow(K+1) up] is one buffer, the
    stride-2 convolution of the next level reads its first segment."""
    from ..plane_graph import PlaneGraph
    g = PlaneGraph(B, dev)
    L = lambda s: (H >> s, W >> s)
    for name, s, chunks in (("in0", 0, 1), ("c0", 0, 2), ("c1a", 1, 2), ("c1", 1, 4), ("c2a", 2, 4), ("cat2", 2, 7), ("c3a", 3, 8), ("cat3", 3, 13),
                            ("c4a", 4, 16), ("cat4", 4, 25), ("c5a", 5, 16), ("cat5", 5, 33), ("c6a", 6, 32), ("c6", 6, 32),
                            ("ic5", 5, 16), ("ic4", 4, 8), ("ic3", 3, 4), ("ic2", 2, 2)):
        g.buffer(name, *L(s), chunks)
    g.input("in0", 6)
    cv = lambda name: (getattr(net, name)[0].weight, getattr(net, name)[0].bias)
    for name, src, dst, stride in (("conv0", ("in0", 0, 1), ("c0", 0), 1), ("conv1", ("c0", 0, 2), ("c1a", 0), 2),
                                   ("conv1_1", ("c1a", 0, 2), ("c1", 0), 1), ("conv2", ("c1", 0, 4), ("c2a", 0), 2),
                                   ("conv2_1", ("c2a", 0, 4), ("cat2", 0), 1), ("conv3", ("cat2", 0, 4), ("c3a", 0), 2),
                                   ("conv3_1", ("c3a", 0, 8), ("cat3", 0), 1), ("conv4", ("cat3", 0, 8), ("c4a", 0), 2),
                                   ("conv4_1", ("c4a", 0, 16), ("cat4", 0), 1), ("conv5", ("cat4", 0, 16), ("c5a", 0), 2),
                                   ("conv5_1", ("c5a", 0, 16), ("cat5", 0), 1), ("conv6", ("cat5", 0, 16), ("c6a", 0), 2),
                                   ("conv6_1", ("c6a", 0, 32), ("c6", 0), 1)):
        g.conv(*cv(name), src, dst, stride=stride)
    g.predict_flow(net.predict_flow6, ("c6", 0, 32), "flow6")
    prev, prev_chunks = "c6", 32
    for lvl, cat, nskip, ndec, ic in ((5, "cat5", 16, 16, "ic5"), (4, "cat4", 16, 8, "ic4"), (3, "cat3", 8, 4, "ic3"), (2, "cat2", 4, 2, "ic2")):
        total = nskip + ndec + 1
        g.up_flow(getattr(net, f"upsampled_flow{lvl + 1}_to_{lvl}"), f"flow{lvl + 1}", (cat, nskip + ndec))
        g.deconv(*cv(f"deconv{lvl}"), (prev, 0, prev_chunks), (cat, nskip))
        g.conv(*cv(f"inter_conv{lvl}"), (cat, 0, total), (ic, 0), slope=1.0)
        g.predict_flow(getattr(net, f"predict_flow{lvl}"), (ic, 0, g.bufs[ic].chunks), f"flow{lvl}")
        prev, prev_chunks = cat, total
    g.output("flow2")
    return g.build()


def _fusion_graph(net, B, H, W, dev):
    """FlowNetFusion (models/flownet2/FlowNetFusion.py:12-71) at full resolution: cat1 = [conv1_1 128 | deconv1 32 | up 2],
    cat0 = [conv0 64 | deconv0 16 (half a chunk) | up 2] -- the 82 reference channels sit at 0..79 and 96, 97."""
    from ..plane_graph import PlaneGraph
    g = PlaneGraph(B, dev)
    L = lambda s: (H >> s, W >> s)
    for name, s, chunks in (("in0", 0, 1), ("cat0", 0, 4), ("c1a", 1, 2), ("cat1", 1, 6), ("c2a", 2, 4), ("c2", 2, 4), ("ic1", 1, 1), ("ic0", 0, 1)):
        g.buffer(name, *L(s), chunks)
    g.input("in0", 11)
    cv = lambda name: (getattr(net, name)[0].weight, getattr(net, name)[0].bias)
    g.conv(*cv("conv0"), ("in0", 0, 1), ("cat0", 0))
    g.conv(*cv("conv1"), ("cat0", 0, 2), ("c1a", 0), stride=2)
    g.conv(*cv("conv1_1"), ("c1a", 0, 2), ("cat1", 0))
    g.conv(*cv("conv2"), ("cat1", 0, 4), ("c2a", 0), stride=2)
    g.conv(*cv("conv2_1"), ("c2a", 0, 4), ("c2", 0))
    g.predict_flow(net.predict_flow2, ("c2", 0, 4), "flow2")
    g.up_flow(net.upsampled_flow2_to_1, "flow2", ("cat1", 5))
    g.deconv(*cv("deconv1"), ("c2", 0, 4), ("cat1", 4))
    g.conv(*cv("inter_conv1"), ("cat1", 0, 6), ("ic1", 0), slope=1.0)
    g.predict_flow(net.predict_flow1, ("ic1", 0, 1), "flow1")
    g.up_flow(net.upsampled_flow1_to_0, "flow1", ("cat0", 3))
    g.deconv(*cv("deconv0"), ("cat1", 0, 6), ("cat0", 2))
    seg0 = [(0, 64, 0), (64, 16, 64), (80, 2, 96)]
    g.conv(*cv("inter_conv0"), ("cat0", 0, 4), ("ic0", 0), slope=1.0, in_segments=seg0)
    g.predict_flow(net.predict_flow0, ("ic0", 0, 1), "flow0")
    g.output("flow0")
    return g.build()


_BRANCH_STREAMS: dict = {}


def _branch_stream(device):
    key = torch.device(device)
    if key not in _BRANCH_STREAMS:
        _BRANCH_STREAMS[key] = torch.cuda.Stream(device=key)
    return _BRANCH_STREAMS[key]


class FlowNet2(nn.Module):
    def __init__(self, batchNorm=False, div_flow=20.0, return_feat_maps: bool = False):
        super().__init__()
        self.div_flow = div_flow
        self.channelnorm = ChannelNorm()
        self.flownetc = FlowNetC(batchNorm=batchNorm)
        self.resample1 = Resample2d()
        self.flownets_1 = FlowNetS(batchNorm=batchNorm)
        self.resample2 = Resample2d()
        self.flownets_2 = FlowNetS(batchNorm=batchNorm)
        self.flownets_d = FlowNetSD(batchNorm=batchNorm)
        self.resample3 = Resample2d()
        self.resample4 = Resample2d()
        self.flownetfusion = FlowNetFusion(batchNorm=batchNorm)
        self.register_buffer("_mean64", torch.tensor(_RGB_MEAN, dtype=torch.float64).view(1, 3, 1, 1),
                             persistent=False)

    def _warp_stage(self, x, flow):
        """flownet2_models.py:138-145: warp frame 2 by `flow`, brightness error and its channel norm."""
        resampled = self.resample1(x[:, 3:], flow)
        diff = x[:, :3] - resampled
        return torch.cat((x, resampled, flow / self.div_flow, self.channelnorm(diff)), dim=1)

    def _forward_fused(self, x1, x2):
        """The native path (frozen parameters, HIP float32, sides multiples of 64): the sub-networks strung together by the fused
        Functions of fn2_glue.py / csrc/fn2_glue.hip instead of ~20 torch operators per stage, FlowNet-SD on a second HIP stream."""
        from ..fn2_glue import fusion_input, normalize_pair, upscale4, warp_stage
        x = normalize_pair(x1, x2, self._mean64.reshape(-1))               # :93-96, :124-125 in one pass (bit-exact)
        fork = os.environ.get("UFR_FN2_BRANCH_STREAM", "1") != "0"
        main, side = torch.cuda.current_stream(x.device), _branch_stream(x.device)
        # (FlowNet-SD and FlowNetC differentiate `x` itself, which has five consumers: their input gradients are handed over as clones)
        def sd():
            with _L.static_handoff(input_grads=False):
                return upscale4(self.flownets_d(x)[0], False, self.div_flow, divide=True)    # sic: divided (:176); nearest x4
        # every sub-network's flow is read at once by `upscale4` / the next stage and kept by nobody: the engines hand over aliases
        # of their static buffers instead of clones (`_lib.static_handoff`); the fusion network's result goes to the caller and is
        # cloned as ever
        with _L.static_handoff():
            if fork:                                                       # (why a second stream: see forward())
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    flow_sd = sd()
            with _L.static_handoff(input_grads=False):
                flow_c = upscale4(self.flownetc(x)[0], True, self.div_flow)    # == interpolate(flow * div_flow, x4, bilinear)
            flow_s1 = upscale4(self.flownets_1(warp_stage(x, flow_c, self.div_flow))[0], True, self.div_flow)
            flow_s2 = upscale4(self.flownets_2(warp_stage(x, flow_s1, self.div_flow))[0], False, self.div_flow)
            if fork:
                main.wait_stream(side)
                flow_sd.record_stream(main)
            else:
                flow_sd = sd()
        return self.flownetfusion(fusion_input(x, flow_sd, flow_s2))

    def forward(self, x1, x2):
        if (x1.is_cuda and x1.dtype == torch.float32 and _L.engine_refusal(self, x1, 64) is None
                and os.environ.get("UFR_FN2_GLUE", "1") != "0"):
            return self._forward_fused(x1, x2)
        # the torch spelling of flownet2_models.py:122-205 (UFR_FN2_GLUE=0 keeps it on the native path too: the yardstick of
        # tests/test_fn2_glue_gpu.py)
        x1 = (x1.double() - self._mean64).float()                  # :93-96, :124-125
        x2 = (x2.double() - self._mean64).float()
        x = torch.cat((x1, x2), dim=1)
        up_bl = lambda f: F.interpolate(f, scale_factor=4, mode="bilinear", align_corners=False)
        up_nn = lambda f: F.interpolate(f, scale_factor=4, mode="nearest")

        def small_displacement_branch():
            flow_sd = up_nn(self.flownets_d(x)[0] / self.div_flow)     # sic: divided (:176)
            return flow_sd, self.channelnorm(flow_sd), self.channelnorm(x[:, :3] - self.resample3(x[:, 3:], flow_sd))

        # FlowNet-SD reads the frames only (flownet2_models.py:174-181): it does not depend on the FlowNetC -> FlowNetS -> FlowNetS
        # chain, and at one pair per GPU most launches of either cover a fraction of the 256 CUs (1/16 .. 1/64 grids).  On the
        # native path the branch runs on a second HIP stream -- forward here, its adjoint wherever autograd replays these nodes
        # (the stream they were recorded on) -- and the two meet again at the fusion network's input.  Inside a HIP-graph capture
        # the fork / join become graph edges.  UFR_FN2_BRANCH_STREAM=0: one stream (the A/B switch).
        fork = x.is_cuda and _L.engine_refusal(self.flownets_d, x, 64) is None and os.environ.get("UFR_FN2_BRANCH_STREAM", "1") != "0"
        if fork:
            main, side = torch.cuda.current_stream(x.device), _branch_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                flow_sd, norm_sd, err_sd = small_displacement_branch()

        flow_c = up_bl(self.flownetc(x)[0] * self.div_flow)
        flow_s1 = up_bl(self.flownets_1(self._warp_stage(x, flow_c))[0] * self.div_flow)
        flow_s2 = up_nn(self.flownets_2(self._warp_stage(x, flow_s1))[0] * self.div_flow)
        norm_s2 = self.channelnorm(flow_s2)
        err_s2 = self.channelnorm(x[:, :3] - self.resample4(x[:, 3:], flow_s2))

        if fork:
            main.wait_stream(side)
            for t_ in (flow_sd, norm_sd, err_sd):
                t_.record_stream(main)
        else:
            flow_sd, norm_sd, err_sd = small_displacement_branch()

        fused_in = torch.cat((x[:, :3], flow_sd, flow_s2, norm_sd, norm_s2, err_sd, err_s2), dim=1)
        return self.flownetfusion(fused_in)

"""Column-band data gradients for the convolutions behind a windowed prefix (patch_attack.py, cone.py).

When FlowNetC's conv1-3 run on a window around the patch, the rest of the network still needs its adjoint
only where activations depend on that window.  Behind the 21x21 stride-2 correlation that is every row but
less than half of the columns for conv3_1 / conv4 / conv4_1 / conv5 (the most expensive data gradients of
the head).  `band_conv2d` is `F.conv2d` whose backward computes the input gradient on a column band only:
    gather the band of grad_output (origin in device memory: one captured graph serves every placement)
    -> MIOpen backward-data on the band -> scatter the exact columns into a zeroed full-size gradient.
Columns within `kernel - 1 - padding` cells of an interior band edge miss contributions from outside the
band and are left zero; the band is sized (patch_attack.py) so that they are never needed.  Weights get no
gradient: the attack differentiates with respect to the frames only.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib as L


class Band:
    """A per-sample column band: `win` int32 [B,8] with win[:,1] = first pixel column (win[:,0] = 0), `width`
    pixels wide; both multiples of every level stride the band is used at."""

    def __init__(self, win: torch.Tensor, width: int, cone_win: torch.Tensor | None = None, cone_hw=None):
        self.win, self.width = win, int(width)
        # the prefix window itself (pixels): the correlation's adjoint is needed on its cells only
        self.cone_win, self.cone_hw = cone_win, cone_hw
        # incremental forward (second and later iterations of an attack() call): the blocks named in
        # `inc_layers` recompute the band's columns only and paste them into `caches[name]`, the activations of
        # the previous iteration; `incremental` is switched by the step between its two captured graphs
        self.inc_layers, self.incremental, self.caches = (), False, {}


class _WindowCorrelation(torch.autograd.Function):
    """spatial_correlation_sample(kernel 1, stride 1, padding 0) whose backward fills only the cells of the
    prefix window (csrc/correlation_window.hip); everything else of both input gradients is zero."""

    @staticmethod
    def forward(ctx, input1, input2, patch, dilation_patch, band, in_stride):
        from . import spatial_correlation_sampler_backend as correlation
        ctx.save_for_backward(input1, input2)
        ctx.meta = (int(patch), int(dilation_patch), band, int(in_stride))
        return correlation.forward(input1, input2, 1, 1, patch, patch, 0, 0, 1, 1, dilation_patch, dilation_patch, 1, 1)

    @staticmethod
    def backward(ctx, gout):
        input1, input2 = ctx.saved_tensors
        patch, dil, band, ls = ctx.meta
        B, Cn, H, W = input1.shape
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        L.check(L.lib().ufr_corr_backward_window(L.ptr(input1), L.ptr(input2), L.ptr(gout.contiguous()), L.ptr(g1), L.ptr(g2),
                                                 B, Cn, H, W, patch, dil, L.ptr(band.cone_win), ls, band.cone_hw[0] // ls,
                                                 band.cone_hw[1] // ls, L.stream()), "corr backward window")
        return g1, g2, None, None, None, None


def window_correlation(input1, input2, patch, dilation_patch, band: Band, in_stride: int):
    """Cost volume [B,P,P,H,W] of two feature maps that vary only inside `band.cone_win`."""
    if input1.dtype != torch.float32 or not input1.is_contiguous() or not input2.is_contiguous():
        raise TypeError("window_correlation: contiguous float32 feature maps")
    return _WindowCorrelation.apply(input1, input2, patch, dilation_patch, band, in_stride)


def _conv_forward(x, weight, s, p):
    """conv2d without bias inside the band Functions (no autograd here)."""
    return F.conv2d(x, weight, None, s, p)


class _BandConv2d(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, stride, padding, band, in_stride):
        ctx.save_for_backward(weight)
        ctx.meta = (tuple(x.shape), int(stride), int(padding), band, int(in_stride))
        return F.conv2d(x, weight, bias, stride, padding)

    @staticmethod
    def backward(ctx, gy):
        (weight,) = ctx.saved_tensors
        in_shape, s, p, band, ls_in = ctx.meta
        return _band_data_gradient(gy, weight, in_shape, s, p, band, ls_in), None, None, None, None, None, None


def _band_data_gradient(gy, weight, in_shape, s, p, band, ls_in):
    """Input gradient of conv2d(weight, stride s, padding p) on the band's columns, zero elsewhere."""
    B, Cin, Hi, Wi = in_shape
    k = weight.shape[-1]
    ls_out = ls_in * s
    wib, wob = band.width // ls_in, band.width // ls_out
    _, Cout, Ho, Wo = gy.shape
    lib, st = L.lib(), L.stream()
    gy = gy.contiguous()
    gyb = torch.empty(B, Cout, Ho, wob, dtype=gy.dtype, device=gy.device)
    L.check(lib.ufr_window_gather(L.ptr(gy), L.ptr(gyb), L.ptr(band.win), B, B, Cout, Ho, Wo, Ho, wob, ls_out, 0, st),
            "band gather")
    gxb = torch.ops.aten.convolution_backward(gyb, gyb.new_empty((B, Cin, Hi, wib)), weight, None, (s, s), (p, p),
                                              (1, 1), False, (0, 0), 1, (True, False, False))[0]
    gx = torch.zeros(B, Cin, Hi, Wi, dtype=gy.dtype, device=gy.device)
    L.check(lib.ufr_window_scatter(L.ptr(gxb.contiguous()), L.ptr(gx), L.ptr(band.win), B, B, Cin, Hi, Wi, Hi, wib,
                                   ls_in, k - 1 - p, st), "band scatter")
    return gx


class _IncrementalConvLeaky(torch.autograd.Function):
    """A conv + bias + LeakyReLU block whose input differs from the previous iteration's only inside the band:
    gather the band of x, convolve, apply the epilogue, paste the exact columns into `cache` (the previous
    iteration's activations, updated in place and returned).  Backward = the banded data gradient."""

    @staticmethod
    def forward(ctx, x, cache, weight, bias, stride, padding, slope, band, in_stride):
        B, Cin, Hi, Wi = x.shape
        s, p, k = int(stride), int(padding), weight.shape[-1]
        ls_out = in_stride * s
        wib, wob = band.width // in_stride, band.width // ls_out
        _, Cout, Ho, Wo = cache.shape
        lib, st = L.lib(), L.stream()
        xb = torch.empty(B, Cin, Hi, wib, dtype=x.dtype, device=x.device)
        L.check(lib.ufr_window_gather(L.ptr(x), L.ptr(xb), L.ptr(band.win), B, B, Cin, Hi, Wi, Hi, wib, in_stride, 0, st),
                "band gather")
        yb = _conv_forward(xb, weight, s, p).contiguous()
        L.check(lib.ufr_bias_leaky_forward(L.ptr(yb), L.ptr(bias), B, Cout, Ho * wob, float(slope), st), "bias leaky")
        L.check(lib.ufr_window_scatter(L.ptr(yb), L.ptr(cache), L.ptr(band.win), B, B, Cout, Ho, Wo, Ho, wob, ls_out,
                                       k - 1 - p, st), "band paste")
        ctx.mark_dirty(cache)
        ctx.save_for_backward(weight, cache)
        ctx.meta = ((B, Cin, Hi, Wi), s, p, float(slope), band, int(in_stride))
        return cache

    @staticmethod
    def backward(ctx, gy):
        weight, y = ctx.saved_tensors
        in_shape, s, p, slope, band, ls_in = ctx.meta
        gy = gy.contiguous()
        g_pre = torch.empty_like(gy)
        L.check(L.lib().ufr_leaky_backward(L.ptr(y), L.ptr(gy), L.ptr(g_pre), gy.numel(), slope, L.stream()),
                "leaky backward")
        return _band_data_gradient(g_pre, weight, in_shape, s, p, band, ls_in), None, None, None, None, None, None, None, None


class _BiasLeaky(torch.autograd.Function):
    """y = LeakyReLU(x + bias[c]) in place on x (csrc/bias_act.hip); no gradient for the frozen bias."""

    @staticmethod
    def forward(ctx, x, bias, slope):
        B, Cn = x.shape[0], x.shape[1]
        L.check(L.lib().ufr_bias_leaky_forward(L.ptr(x), L.ptr(bias), B, Cn, x.numel() // (B * Cn), float(slope),
                                               L.stream()), "bias leaky forward")
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        ctx.slope = float(slope)
        return x

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gy = gy.contiguous()
        gx = torch.empty_like(gy)
        L.check(L.lib().ufr_leaky_backward(L.ptr(y), L.ptr(gy), L.ptr(gx), gy.numel(), ctx.slope, L.stream()),
                "leaky backward")
        return gx, None, None


class _Conv3x3C2(torch.autograd.Function):
    """Conv2d(Cin, 2, 3, 1, 1) with frozen parameters (csrc/small_cout.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, Cin, H, W = x.shape
        y = torch.empty(B, 2, H, W, dtype=x.dtype, device=x.device)
        n_ws = L.lib().ufr_conv3x3_c2_workspace_floats(B, Cin, H, W)
        ws = torch.empty(n_ws, dtype=x.dtype, device=x.device) if n_ws else None
        L.check(L.lib().ufr_conv3x3_c2_forward(L.ptr(x), L.ptr(weight), L.ptr(bias), L.ptr(y), L.ptr(ws) if n_ws else None,
                                               B, Cin, H, W, L.stream()), "conv3x3 c2 forward")
        ctx.save_for_backward(weight)
        ctx.shape = (B, Cin, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (weight,) = ctx.saved_tensors
        B, Cin, H, W = ctx.shape
        gx = torch.empty(B, Cin, H, W, dtype=gy.dtype, device=gy.device)
        L.check(L.lib().ufr_conv3x3_c2_backward_data(L.ptr(gy.contiguous()), L.ptr(weight), L.ptr(gx), B, Cin, H, W,
                                                     L.stream()), "conv3x3 c2 backward")
        return gx, None, None


class _Deconv4x4C2(torch.autograd.Function):
    """ConvTranspose2d(2, 2, 4, 2, 1) with frozen parameters (csrc/small_cout.hip)."""

    @staticmethod
    def forward(ctx, x, weight, bias):
        B, _, H, W = x.shape
        y = torch.empty(B, 2, 2 * H, 2 * W, dtype=x.dtype, device=x.device)
        L.check(L.lib().ufr_deconv4x4s2_c2_forward(L.ptr(x), L.ptr(weight), L.ptr(bias) if bias is not None else None,
                                                   L.ptr(y), B, H, W, L.stream()), "deconv4x4s2 c2 forward")
        ctx.save_for_backward(weight)
        ctx.shape = (B, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        (weight,) = ctx.saved_tensors
        B, H, W = ctx.shape
        gx = torch.empty(B, 2, H, W, dtype=gy.dtype, device=gy.device)
        L.check(L.lib().ufr_deconv4x4s2_c2_backward_data(L.ptr(gy.contiguous()), L.ptr(weight), L.ptr(gx), B, H, W,
                                                         L.stream()), "deconv4x4s2 c2 backward")
        return gx, None, None


def _frozen(mod):
    return not (mod.weight.requires_grad or (mod.bias is not None and mod.bias.requires_grad))


def flow_head(x, conv: torch.nn.Conv2d):
    """`predict_flow*` (Conv2d(Cin, 2, 3, 1, 1), models/submodules.py:85-86): one pass over x on the device when
    the parameters are frozen (or autograd is off); the module itself otherwise."""
    ok = (x.is_cuda and x.dtype == torch.float32 and conv.out_channels == 2 and conv.kernel_size == (3, 3)
          and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
          and conv.bias is not None and (_frozen(conv) or not torch.is_grad_enabled()))
    if ok:
        return _Conv3x3C2.apply(x.contiguous(), conv.weight, conv.bias)
    return F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)


def flow_upsample(x, deconv: torch.nn.ConvTranspose2d):
    """`upsampled_flow*_to_*` (ConvTranspose2d(2, 2, 4, 2, 1), models/submodules.py:89-90), same conditions."""
    ok = (x.is_cuda and x.dtype == torch.float32 and deconv.in_channels == 2 and deconv.out_channels == 2
          and deconv.kernel_size == (4, 4) and deconv.stride == (2, 2) and deconv.padding == (1, 1)
          and deconv.output_padding == (0, 0) and deconv.dilation == (1, 1) and deconv.groups == 1
          and (_frozen(deconv) or not torch.is_grad_enabled()))
    if ok:
        return _Deconv4x4C2.apply(x.contiguous(), deconv.weight, deconv.bias)
    return F.conv_transpose2d(x, deconv.weight, deconv.bias, deconv.stride, deconv.padding, deconv.output_padding,
                              deconv.groups, deconv.dilation)


class FlowHead(torch.nn.Conv2d):
    """`predict_flow*`: a Conv2d (same parameter names, so checkpoints load unchanged) that runs `flow_head`."""

    def forward(self, x):
        return flow_head(x, self)


class FlowUpsample(torch.nn.ConvTranspose2d):
    """`upsampled_flow*` / PWC-Net's `deconv*`: a ConvTranspose2d that runs `flow_upsample`."""

    def forward(self, x, output_size=None):
        return flow_upsample(x, self)


def conv_relu(x, conv: torch.nn.Conv2d):
    """`F.relu(conv(x))` of RAFT's motion encoder and heads (models/raft/update.py): convolution without bias,
    then bias + ReLU as one in-place pass, under the same conditions as `conv_leaky`."""
    if (x.is_cuda and x.dtype == torch.float32 and conv.bias is not None
            and (_frozen(conv) or not torch.is_grad_enabled())):
        y = band_conv2d(x, conv, None, 0, with_bias=False)
        return _BiasLeaky.apply(y.contiguous(), conv.bias, 0.0)
    return F.relu(conv(x))


def conv_leaky(x, seq, band: Band | None = None, in_stride: int = 0, name: str | None = None):
    """The reference's `conv` / `deconv` block (models/submodules.py:18-46, :75-82) = Sequential(Conv2d or
    ConvTranspose2d with bias, LeakyReLU): on a HIP float32 tensor the convolution runs without bias and
    bias + activation are one in-place pass; with frozen parameters only.  Otherwise the plain modules.
    A block listed in `band.inc_layers` keeps its activations in `band.caches[name]` and, when
    `band.incremental` is set, recomputes the band's columns only."""
    conv, act = seq[0], seq[1]
    frozen = not (conv.weight.requires_grad or (conv.bias is not None and conv.bias.requires_grad))
    fused = (x.is_cuda and x.dtype == torch.float32 and conv.bias is not None and act.negative_slope > 0
             and (frozen or not torch.is_grad_enabled()))
    if not fused:
        return act(band_conv2d(x, conv, band, in_stride)) if isinstance(conv, torch.nn.Conv2d) else act(conv(x))
    tracked = (band is not None and band.width and name in band.inc_layers and x.requires_grad
               and isinstance(conv, torch.nn.Conv2d))
    if tracked and band.incremental:
        # a detached alias: the Function marks its cache argument dirty, the stored tensor itself stays a plain buffer
        return _IncrementalConvLeaky.apply(x.contiguous(), band.caches[name].detach(), conv.weight, conv.bias, conv.stride[0],
                                           conv.padding[0], act.negative_slope, band, in_stride)
    if isinstance(conv, torch.nn.Conv2d):
        y = band_conv2d(x, conv, band, in_stride, with_bias=False)
    else:
        y = F.conv_transpose2d(x, conv.weight, None, conv.stride, conv.padding, conv.output_padding, conv.groups,
                               conv.dilation)
    if not y.is_contiguous():
        y = y.contiguous()
    y = _BiasLeaky.apply(y, conv.bias, act.negative_slope)
    if tracked:                                   # first iteration of a call: remember the activations
        with torch.no_grad():
            cache = band.caches.get(name)
            if cache is None:
                cache = band.caches[name] = torch.empty_like(y)
            cache.copy_(y)
    return y


def band_conv2d(x, conv: torch.nn.Conv2d, band: Band | None, in_stride: int, with_bias: bool = True):
    """`conv(x)`; with a band, the data gradient is computed on the band's columns only.  `in_stride` = pixels
    per cell of x."""
    bias = conv.bias if with_bias else None
    if band is None or not band.width or not x.requires_grad:
        return F.conv2d(x, conv.weight, bias, conv.stride, conv.padding, conv.dilation, conv.groups)
    s, p = conv.stride[0], conv.padding[0]
    if conv.stride[0] != conv.stride[1] or conv.padding[0] != conv.padding[1] or conv.dilation != (1, 1) or conv.groups != 1:
        raise NotImplementedError("band_conv2d: square stride / padding, no dilation or groups")
    if band.width % (in_stride * s) or x.shape[-1] * in_stride < band.width:
        raise ValueError("band width must be a multiple of the output cell size and fit the frame")
    return _BandConv2d.apply(x, conv.weight, bias, s, p, band, in_stride)

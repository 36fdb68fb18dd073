"""ctypes door onto oracle/libufr_oracle.so and oracle/_ref/libufr_corr_ref.so.

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module.  It operates on CPU torch
tensors (viewed as raw pointers) and exposes, for the model-level oracle,
torch.autograd Functions whose forward/backward are the C restatements.

Function names mirror the reference's Python-visible API:
  spatial_correlation_sampler_backend.forward/backward
      (correlation_sampler.cpp:59-124), alt_cuda_corr.forward/backward
      (alt_cuda_corr/correlation.cpp:23-48), resample2d_cuda / channelnorm_cuda.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_REF = None
_f32p, _f64p, _i32 = C.POINTER(C.c_float), C.POINTER(C.c_double), C.c_int


def build(ref: bool = True) -> None:
    """Compile the C restatement (and, when /root/reference is mounted, the reference lib)."""
    targets = ["all"] + (["ref"] if ref else [])
    subprocess.run(["make", "-s", "-C", _HERE] + targets, check=True)


def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get("UFR_ORACLE_LIB") or os.path.join(_HERE, "libufr_oracle.so")   # `make sanitize`
        if not os.path.exists(path):
            build(ref=False)
        _LIB = C.CDLL(path)
    return _LIB


def ref_lib():
    """The reference's own correlation.cpp behind a C ABI, or None if it was never built."""
    global _REF
    if _REF is None:
        path = os.path.join(_HERE, "_ref", "libufr_corr_ref.so")
        if not os.path.exists(path):
            return None
        _REF = C.CDLL(path)
    return _REF


def _p(t: torch.Tensor):
    assert t.device.type == "cpu" and t.is_contiguous(), "oracle works on contiguous CPU tensors"
    return C.c_void_p(t.data_ptr())


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else (int(v[0]), int(v[1]))


def corr_out_hw(H, W, kH, kW, padH, padW, dilH, dilW, dH, dW):
    oH = (H + 2 * padH - ((kH - 1) * dilH + 1)) // dH + 1
    oW = (W + 2 * padW - ((kW - 1) * dilW + 1)) // dW + 1
    return oH, oW


# ----------------------------------------------------------------------------- correlation
def corr_forward(in1, in2, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW,
                 use_ref: bool = False):
    in1, in2 = in1.contiguous(), in2.contiguous()
    B, Cc, H, W = in1.shape
    oH, oW = corr_out_hw(H, W, kH, kW, padH, padW, dilH, dilW, dH, dW)
    out = torch.empty(B, patchH, patchW, oH, oW, dtype=in1.dtype)
    sfx = {torch.float32: "f32", torch.float64: "f64"}[in1.dtype]
    if use_ref:
        prm = (C.c_int * 12)(kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW)
        getattr(ref_lib(), f"ufr_ref_corr_forward_{sfx}")(_p(in1), _p(in2), _p(out), B, Cc, H, W, prm)
    else:
        getattr(lib(), f"ufr_oracle_corr_forward_{sfx}")(
            _p(in1), _p(in2), _p(out), B, Cc, H, W, kH, kW, patchH, patchW, padH, padW, dilH, dilW,
            dpH, dpW, dH, dW)
    return out


def corr_backward(in1, in2, gout, kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW,
                  use_ref: bool = False):
    in1, in2, gout = in1.contiguous(), in2.contiguous(), gout.contiguous()
    B, Cc, H, W = in1.shape
    oH, oW = gout.shape[3], gout.shape[4]
    g1, g2 = torch.empty_like(in1), torch.empty_like(in2)
    sfx = {torch.float32: "f32", torch.float64: "f64"}[in1.dtype]
    if use_ref:
        prm = (C.c_int * 12)(kH, kW, patchH, patchW, padH, padW, dilH, dilW, dpH, dpW, dH, dW)
        getattr(ref_lib(), f"ufr_ref_corr_backward_{sfx}")(
            _p(in1), _p(in2), _p(gout), _p(g1), _p(g2), B, Cc, H, W, oH, oW, prm)
    else:
        getattr(lib(), f"ufr_oracle_corr_backward_{sfx}")(
            _p(in1), _p(in2), _p(gout), _p(g1), _p(g2), B, Cc, H, W, oH, oW, kH, kW, patchH, patchW,
            padH, padW, dilH, dilW, dpH, dpW, dH, dW)
    return g1, g2


class SpatialCorrelationSamplerFunction(torch.autograd.Function):
    """CPU twin of spatial_correlation_sampler.py:46-116 running on the C oracle."""

    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0, dilation=1,
                dilation_patch=1):
        ctx.save_for_backward(input1, input2)
        ctx.prm = (*_pair(kernel_size), *_pair(patch_size), *_pair(padding), *_pair(dilation),
                   *_pair(dilation_patch), *_pair(stride))
        return corr_forward(input1, input2, *ctx.prm)

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g1, g2 = corr_backward(input1, input2, grad_output, *ctx.prm)
        return g1, g2, None, None, None, None, None, None


def spatial_correlation_sample(input1, input2, kernel_size=1, patch_size=1, stride=1, padding=0,
                               dilation=1, dilation_patch=1):
    return SpatialCorrelationSamplerFunction.apply(input1, input2, kernel_size, patch_size, stride,
                                                   padding, dilation, dilation_patch)


# ----------------------------------------------------------------------------- alt_corr
def altcorr_forward(fmap1, fmap2, coords, radius):
    fmap1, fmap2, coords = fmap1.contiguous(), fmap2.contiguous(), coords.contiguous()
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * radius + 1
    corr = torch.empty(B, N, rd * rd, H1, W1, dtype=torch.float32)
    lib().ufr_oracle_altcorr_forward_f32(_p(fmap1), _p(fmap2), _p(coords), _p(corr), B, N, H1, W1,
                                         H2, W2, Cc, radius)
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    fmap1, fmap2, coords = fmap1.contiguous(), fmap2.contiguous(), coords.contiguous()
    corr_grad = corr_grad.contiguous()
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    g1, g2, gc = torch.empty_like(fmap1), torch.empty_like(fmap2), torch.empty_like(coords)
    lib().ufr_oracle_altcorr_backward_f32(_p(fmap1), _p(fmap2), _p(coords), _p(corr_grad), _p(g1),
                                          _p(g2), _p(gc), B, N, H1, W1, H2, W2, Cc, radius)
    return [g1, g2, gc]


# ----------------------------------------------------------------------------- resample2d
def resample2d_forward(input1, input2, output, kernel_size, bilinear):
    B, Cc, Hi, Wi = input1.shape
    _, _, H, W = input2.shape
    lib().ufr_oracle_resample2d_forward_f32(_p(input1), _p(input2), _p(output), B, Cc, Hi, Wi, H, W,
                                            int(kernel_size), int(bool(bilinear)))
    return 1


def resample2d_backward(input1, input2, grad_output, grad_input1, grad_input2, kernel_size, bilinear):
    B, Cc, Hi, Wi = input1.shape
    _, _, H, W = input2.shape
    lib().ufr_oracle_resample2d_backward_f32(_p(input1), _p(input2), _p(grad_output),
                                             _p(grad_input1), _p(grad_input2), B, Cc, Hi, Wi, H, W,
                                             int(kernel_size), int(bool(bilinear)))
    return 1


class Resample2dFunction(torch.autograd.Function):
    """CPU twin of resample2d.py:7-45."""

    @staticmethod
    def forward(ctx, input1, input2, kernel_size=1, bilinear=True):
        input1, input2 = input1.contiguous(), input2.contiguous()
        ctx.save_for_backward(input1, input2)
        ctx.k, ctx.bl = kernel_size, bilinear
        out = input1.new_zeros(input2.shape[0], input1.shape[1], input2.shape[2], input2.shape[3])
        resample2d_forward(input1, input2, out, kernel_size, bilinear)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input1, input2 = ctx.saved_tensors
        g1, g2 = torch.zeros_like(input1), torch.zeros_like(input2)
        resample2d_backward(input1, input2, grad_output.contiguous(), g1, g2, ctx.k, ctx.bl)
        return g1, g2, None, None


# ----------------------------------------------------------------------------- channelnorm
def channelnorm_forward(input1, output, norm_deg=2):
    B, Cc, H, W = input1.shape
    lib().ufr_oracle_channelnorm_forward_f32(_p(input1), _p(output), B, Cc, H, W)
    return 1


def channelnorm_backward(input1, output, grad_output, grad_input1, norm_deg=2):
    B, Cc, H, W = input1.shape
    lib().ufr_oracle_channelnorm_backward_f32(_p(input1), _p(output), _p(grad_output),
                                              _p(grad_input1), B, Cc, H, W)
    return 1


class ChannelNormFunction(torch.autograd.Function):
    """CPU twin of channelnorm.py:6-29."""

    @staticmethod
    def forward(ctx, input1, norm_deg=2):
        input1 = input1.contiguous()
        out = input1.new_zeros(input1.shape[0], 1, input1.shape[2], input1.shape[3])
        channelnorm_forward(input1, out, norm_deg)
        ctx.save_for_backward(input1, out)
        return out

    @staticmethod
    def backward(ctx, grad_output):
        input1, out = ctx.saved_tensors
        g = torch.zeros_like(input1)
        channelnorm_backward(input1, out, grad_output.contiguous(), g)
        return g, None


# ----------------------------------------------------------------------------- RAFT lookup
def corr_lookup(pyramid, coords, radius):
    """pyramid: list of [B*H1*W1,1,Hl,Wl]; coords: [B,2,H1,W1] -> [B, L*(2r+1)^2, H1, W1]."""
    L = len(pyramid)
    pyramid = [p.contiguous() for p in pyramid]
    coords = coords.contiguous()
    B, _, H1, W1 = coords.shape
    rd = 2 * radius + 1
    out = torch.empty(B, L * rd * rd, H1, W1, dtype=torch.float32)
    ptrs = (C.c_void_p * L)(*[p.data_ptr() for p in pyramid])
    Hl = (C.c_int * L)(*[p.shape[-2] for p in pyramid])
    Wl = (C.c_int * L)(*[p.shape[-1] for p in pyramid])
    lib().ufr_oracle_corr_lookup_f32(ptrs, Hl, Wl, L, _p(coords), _p(out), B, H1, W1, radius)
    return out

"""Drop-in for the reference's pybind module `spatial_correlation_sampler_backend`
(models/Pytorch-Correlation-extension/Correlation_Module/correlation_sampler.cpp:59-129):
same function names, argument order and error behaviour, running the gfx950 kernels of
csrc/correlation.hip through the C ABI.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def _params(kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
            dilation_patchW, dH, dW) -> L.CorrParams:
    return L.CorrParams(int(kH), int(kW), int(patchH), int(patchW), int(padH), int(padW),
                        int(dilationH), int(dilationW), int(dilation_patchH), int(dilation_patchW),
                        int(dH), int(dW))


def _out_hw(H, W, p: L.CorrParams):
    oH = (H + 2 * p.padH - ((p.kH - 1) * p.dilationH + 1)) // p.dH + 1
    oW = (W + 2 * p.padW - ((p.kW - 1) * p.dilationW + 1)) // p.dW + 1
    return oH, oW


def _check_pair(input1, input2):
    L.require_hip(input1, "input1")
    L.require_hip(input2, "input2")
    if input1.device != input2.device:
        raise RuntimeError("input1 is not on same device as input2")
    if input1.shape != input2.shape or input1.dim() != 4:
        raise RuntimeError(f"input1 {tuple(input1.shape)} and input2 {tuple(input2.shape)} must be equal 4-D shapes")
    if input1.dtype != input2.dtype:
        raise RuntimeError("input1 and input2 must have the same dtype")


def forward(input1, input2, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW,
            dilation_patchH, dilation_patchW, dH, dW, scale: float = 1.0, slope: float = 1.0):
    """correlation_sampler.cpp:59-87 -> Tensor[B,patchH,patchW,oH,oW].

    `scale` / `slope` (not in the reference signature, default = identity) expose the fused
    `/C` + LeakyReLU epilogue of include/ufr_hip.h::ufr_corr_forward_fused.
    """
    _check_pair(input1, input2)
    p = _params(kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
                dilation_patchW, dH, dW)
    B, Cc, H, W = input1.shape
    oH, oW = _out_hw(H, W, p)
    if oH <= 0 or oW <= 0:
        raise RuntimeError(f"correlation output would be empty ({oH} x {oW})")
    with torch.cuda.device(input1.device):
        out = torch.empty((B, p.patchH, p.patchW, oH, oW), dtype=input1.dtype, device=input1.device)
        L.check(L.lib().ufr_corr_forward_fused(L.ptr(input1), L.ptr(input2), L.ptr(out),
                                               L.dtype_code(input1), B, Cc, H, W, C.byref(p),
                                               float(scale), float(slope), L.stream()),
                "spatial_correlation_sampler_backend.forward")
    return out


def backward(input1, input2, grad_output, kH, kW, patchH, patchW, padH, padW, dilationH, dilationW,
             dilation_patchH, dilation_patchW, dH, dW):
    """correlation_sampler.cpp:89-124 -> [grad_input1, grad_input2]."""
    _check_pair(input1, input2)
    L.require_hip(grad_output, "grad_output", contiguous=False)
    if grad_output.device != input1.device:
        raise RuntimeError("input1 is not on same device as grad_output")
    grad_output = grad_output.contiguous()  # the reference never checks; be safe, not sorry
    p = _params(kH, kW, patchH, patchW, padH, padW, dilationH, dilationW, dilation_patchH,
                dilation_patchW, dH, dW)
    B, Cc, H, W = input1.shape
    oH, oW = _out_hw(H, W, p)
    if tuple(grad_output.shape) != (B, p.patchH, p.patchW, oH, oW):
        raise RuntimeError(f"grad_output has shape {tuple(grad_output.shape)}, expected "
                           f"{(B, p.patchH, p.patchW, oH, oW)}")
    with torch.cuda.device(input1.device):
        g1, g2 = torch.empty_like(input1), torch.empty_like(input2)
        L.check(L.lib().ufr_corr_backward(L.ptr(input1), L.ptr(input2), L.ptr(grad_output), L.ptr(g1),
                                          L.ptr(g2), L.dtype_code(input1), B, Cc, H, W, C.byref(p),
                                          L.stream()),
                "spatial_correlation_sampler_backend.backward")
    return [g1, g2]

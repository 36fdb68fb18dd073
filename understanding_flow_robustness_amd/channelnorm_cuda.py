"""Drop-in for the reference's pybind module `channelnorm_cuda`
(models/channelnorm_package/channelnorm_cuda.cc:6-30): caller-allocated outputs, returns 1,
`norm_deg` accepted and ignored exactly like the reference kernel (channelnorm_kernel.cu:53-59)."""
from __future__ import annotations

import torch

from . import _lib as L


def _check(**tensors):
    for n, t in tensors.items():
        L.require_hip(t, n)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{n} must be float32 on the gfx950 build")
        if t.dim() != 4:
            raise RuntimeError(f"{n} must be 4-D")


def forward(input1, output, norm_deg):
    _check(input1=input1, output=output)
    B, Cc, H, W = input1.shape
    if tuple(output.shape) != (B, 1, H, W):
        raise RuntimeError("channelnorm: output must be [B,1,H,W]")
    with torch.cuda.device(input1.device):
        L.check(L.lib().ufr_channelnorm_forward(L.ptr(input1), L.ptr(output), B, Cc, H, W, int(norm_deg),
                                                L.stream()), "channelnorm_cuda.forward")
    return 1


def backward(input1, output, gradOutput, gradInput1, norm_deg):
    _check(input1=input1, output=output, gradOutput=gradOutput, gradInput1=gradInput1)
    B, Cc, H, W = input1.shape
    if tuple(output.shape) != (B, 1, H, W) or tuple(gradOutput.shape) != (B, 1, H, W) \
            or gradInput1.shape != input1.shape:
        raise RuntimeError("channelnorm backward: buffer shapes do not match")
    with torch.cuda.device(input1.device):
        L.check(L.lib().ufr_channelnorm_backward(L.ptr(input1), L.ptr(output), L.ptr(gradOutput),
                                                 L.ptr(gradInput1), B, Cc, H, W, int(norm_deg), L.stream()),
                "channelnorm_cuda.backward")
    return 1

"""Differentiable FlowNet2 helpers on the gfx950 kernels (csrc/warp_norm.hip).

The public names -- `Resample2d`, `Resample2dFunction`, `ChannelNorm`, `ChannelNormFunction` -- and their call
signatures are the ones FlowNet2's model code uses (models/resample2d_package/resample2d.py,
models/channelnorm_package/channelnorm.py); the two package modules re-export them from here.
Both operators share one small base: allocate the outputs, hand device pointers to the C ABI through the
`resample2d_cuda` / `channelnorm_cuda` mirrors, remember what the adjoint needs.
"""
from __future__ import annotations

import torch
from torch import nn

from . import channelnorm_cuda, resample2d_cuda


class _KernelOp(torch.autograd.Function):
    """Shared plumbing: subclasses provide `_launch_forward(ctx, *inputs) -> output` and
    `_launch_backward(ctx, grad) -> tuple of input gradients` (tensor inputs first)."""

    @classmethod
    def forward(cls, ctx, *args):
        ctx.n_args = len(args)
        tensors = [a.contiguous() if torch.is_tensor(a) else a for a in args]
        return cls._launch_forward(ctx, *tensors)

    @classmethod
    def backward(cls, ctx, grad):
        grads = cls._launch_backward(ctx, grad.contiguous())
        return grads + (None,) * (ctx.n_args - len(grads))


class Resample2dFunction(_KernelOp):
    """Backward warp of `image` by `flow` (bilinear by default), output at the flow's resolution."""

    @staticmethod
    def _launch_forward(ctx, image, flow, kernel_size=1, bilinear=True):
        ctx.opts = (kernel_size, bilinear)
        ctx.save_for_backward(image, flow)
        warped = image.new_empty((flow.shape[0], image.shape[1], flow.shape[2], flow.shape[3]))
        resample2d_cuda.forward(image, flow, warped, kernel_size, bilinear)
        return warped

    @staticmethod
    def _launch_backward(ctx, grad):
        image, flow = ctx.saved_tensors
        g_image, g_flow = torch.empty_like(image), torch.empty_like(flow)     # written completely by the kernels
        resample2d_cuda.backward(image, flow, grad, g_image, g_flow, *ctx.opts)
        return g_image, g_flow


class ChannelNormFunction(_KernelOp):
    """L2 norm over the channel axis, [B,C,H,W] -> [B,1,H,W]."""

    @staticmethod
    def _launch_forward(ctx, x, norm_deg=2):
        ctx.norm_deg = norm_deg
        norm = x.new_empty((x.shape[0], 1, x.shape[2], x.shape[3]))
        channelnorm_cuda.forward(x, norm, norm_deg)
        ctx.save_for_backward(x, norm)
        return norm

    @staticmethod
    def _launch_backward(ctx, grad):
        x, norm = ctx.saved_tensors
        g_x = torch.empty_like(x)
        channelnorm_cuda.backward(x, norm, grad, g_x, ctx.norm_deg)
        return (g_x,)


class Resample2d(nn.Module):
    def __init__(self, kernel_size=1, bilinear=True):
        super().__init__()
        self.kernel_size, self.bilinear = kernel_size, bilinear

    def forward(self, input1, input2):
        return Resample2dFunction.apply(input1, input2, self.kernel_size, self.bilinear)


class ChannelNorm(nn.Module):
    def __init__(self, norm_deg=2):
        super().__init__()
        self.norm_deg = norm_deg

    def forward(self, input1):
        return ChannelNormFunction.apply(input1, self.norm_deg)

"""Host side of csrc/raft_glue.hip: the elementwise work in front of RAFT's encoders and of its on-the-fly correlation as two
autograd Functions (one kernel each way) instead of ~35 torch operators per forward / backward:

    normalize_pair(image1, image2)    `2 * (image / 255.0) - 1.0` for both frames (models/raft/raft.py:128-129), as the stack [2B,3,H,W]
                                      the feature encoder concatenates them to (raft.py:141)
    fmap_pyramid(fmap, levels)        AlternateCorrBlock's pyramid (models/raft/corr.py:97-105, :128-129): the map permuted to NHWC and
                                      `levels - 1` average poolings of it, each permuted to NHWC

Both are bit for bit the torch spelling (tests/test_raft_glue_gpu.py compares with torch.equal), so every golden of the model holds
unchanged; UFR_RAFT_GLUE=0 keeps the torch operators.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L


def enabled(*tensors) -> bool:
    return (os.environ.get("UFR_RAFT_GLUE", "1") != "0" and all(t.is_cuda and t.dtype == torch.float32 for t in tensors))


class _NormalizePair(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image1, image2):
        image1, image2 = image1.contiguous(), image2.contiguous()
        stack = image1.new_empty(2 * image1.shape[0], *image1.shape[1:])
        with torch.cuda.device(image1.device):
            L.check(L.lib().ufr_raft_normalize_pair(L.ptr(image1), L.ptr(image2), L.ptr(stack), image1.numel(), L.stream()), "raft normalize pair")
        ctx.shape = tuple(image1.shape)
        return stack

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        g1 = g.new_empty(ctx.shape)
        g2 = g.new_empty(ctx.shape) if ctx.needs_input_grad[1] else None
        with torch.cuda.device(g.device):
            L.check(L.lib().ufr_raft_normalize_pair_backward(L.ptr(g), L.ptr(g1), L.ptr(g2) if g2 is not None else None, g1.numel(), L.stream()),
                    "raft normalize pair backward")
        return g1, g2


def normalize_pair(image1, image2):
    """[2B,3,H,W]: rows [:B] = 2 * (image1 / 255) - 1, rows [B:] the same of image2; None when the kernels do not serve the tensors."""
    if not enabled(image1, image2) or image1.shape != image2.shape or image1.numel() % 4 or image1.device != image2.device:
        return None
    return _NormalizePair.apply(image1, image2)


def _pointer_row(tensors):
    return (C.c_void_p * len(tensors))(*[t.data_ptr() if t is not None else None for t in tensors])


class _FmapPyramid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmap, levels):
        fmap = fmap.contiguous()
        B, Cn, H, W = fmap.shape
        outs = [fmap.new_empty(B, H >> l, W >> l, Cn) for l in range(levels)]
        with torch.cuda.device(fmap.device):
            L.check(L.lib().ufr_raft_fmap_pyramid_forward(L.ptr(fmap), _pointer_row(outs), levels, B, Cn, H, W, L.stream()), "raft fmap pyramid")
        ctx.meta = (levels, B, Cn, H, W)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        levels, B, Cn, H, W = ctx.meta
        grads = [g.contiguous() if g is not None else None for g in grads]
        ref = next(g for g in grads if g is not None)
        gf = ref.new_empty(B, Cn, H, W)
        with torch.cuda.device(ref.device):
            L.check(L.lib().ufr_raft_fmap_pyramid_backward(_pointer_row(grads), levels, L.ptr(gf), B, Cn, H, W, L.stream()),
                    "raft fmap pyramid backward")
        return gf, None


def fmap_pyramid(fmap, levels: int):
    """(level 0 .. levels - 1) of `fmap` [B,C,H,W] as NHWC tensors, level l = avg_pool2d(level l - 1, 2, stride=2); None when the
    kernels do not serve the tensor."""
    if not enabled(fmap) or not 1 <= levels <= 4 or min(fmap.shape[2] >> (levels - 1), fmap.shape[3] >> (levels - 1)) < 1:
        return None
    return _FmapPyramid.apply(fmap, levels)

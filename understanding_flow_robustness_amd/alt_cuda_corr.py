"""Drop-in for the reference's pybind module `alt_cuda_corr`
(models/alt_cuda_corr/correlation.cpp:23-54): forward / backward with the same argument order,
list-valued returns and input checks, on the gfx950 kernels of csrc/raft_corr.hip.

The reference never wires `backward` into autograd (models/raft/corr.py:132 calls the raw
`forward`), which makes RAFT's alternate-correlation path non-differentiable there.
`AltCorrFunction` below adds the missing autograd wrapper so the attack loop can use it.
"""
from __future__ import annotations

import torch

from . import _lib as L


def _check(fmap1, fmap2, coords, radius):
    for t, n in ((fmap1, "fmap1"), (fmap2, "fmap2"), (coords, "coords")):
        L.require_hip(t, n)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{n} must be float32 (alt_cuda_corr is fp32 only, correlation_kernel.cu:278)")
    if fmap1.dim() != 4 or fmap2.dim() != 4 or coords.dim() != 5 or coords.shape[-1] != 2:
        raise RuntimeError("expected fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords [B,N,H1,W1,2]")
    if fmap1.shape[0] != fmap2.shape[0] or fmap1.shape[3] != fmap2.shape[3]:
        raise RuntimeError("fmap1 and fmap2 disagree in batch or channel size")
    if tuple(coords.shape[2:4]) != tuple(fmap1.shape[1:3]) or coords.shape[0] != fmap1.shape[0]:
        raise RuntimeError("coords must be [B,N,H1,W1,2] matching fmap1")
    if int(radius) < 0:
        raise RuntimeError("radius must be >= 0")


def forward(fmap1, fmap2, coords, radius):
    """correlation.cpp:23-33 -> [corr [B,N,(2r+1)^2,H1,W1]]."""
    _check(fmap1, fmap2, coords, radius)
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * int(radius) + 1
    with torch.cuda.device(fmap1.device):
        corr = torch.empty((B, N, rd * rd, H1, W1), dtype=torch.float32, device=fmap1.device)
        L.check(L.lib().ufr_altcorr_forward(L.ptr(fmap1), L.ptr(fmap2), L.ptr(coords), L.ptr(corr), B, N,
                                            H1, W1, H2, W2, Cc, int(radius), L.stream()),
                "alt_cuda_corr.forward")
    return [corr]


def backward(fmap1, fmap2, coords, corr_grad, radius):
    """correlation.cpp:36-48 -> [fmap1_grad, fmap2_grad, coords_grad (zeros)]."""
    _check(fmap1, fmap2, coords, radius)
    L.require_hip(corr_grad, "corr_grad")
    B, H1, W1, Cc = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    N = coords.shape[1]
    rd = 2 * int(radius) + 1
    if tuple(corr_grad.shape) != (B, N, rd * rd, H1, W1):
        raise RuntimeError(f"corr_grad has shape {tuple(corr_grad.shape)}, expected {(B, N, rd * rd, H1, W1)}")
    with torch.cuda.device(fmap1.device):
        g1, g2, gc = torch.empty_like(fmap1), torch.empty_like(fmap2), torch.empty_like(coords)
        L.check(L.lib().ufr_altcorr_backward(L.ptr(fmap1), L.ptr(fmap2), L.ptr(coords), L.ptr(corr_grad),
                                             L.ptr(g1), L.ptr(g2), L.ptr(gc), B, N, H1, W1, H2, W2, Cc,
                                             int(radius), L.stream()),
                "alt_cuda_corr.backward")
    return [g1, g2, gc]


class AltCorrFunction(torch.autograd.Function):
    """The autograd Function the reference lacks: differentiable alt_corr."""

    @staticmethod
    def forward(ctx, fmap1, fmap2, coords, radius):
        fmap1, fmap2, coords = fmap1.contiguous(), fmap2.contiguous(), coords.contiguous()
        ctx.save_for_backward(fmap1, fmap2, coords)
        ctx.radius = int(radius)
        return forward(fmap1, fmap2, coords, radius)[0]

    @staticmethod
    def backward(ctx, corr_grad):
        fmap1, fmap2, coords = ctx.saved_tensors
        g1, g2, _ = backward(fmap1, fmap2, coords, corr_grad.contiguous(), ctx.radius)
        return g1, g2, None, None  # coords carry no gradient (reference: zeros, and detached anyway)

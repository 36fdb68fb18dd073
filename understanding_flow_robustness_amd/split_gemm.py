"""float32-accurate matrix product on the bf16 matrix cores (csrc/split_gemm.hip): the measured
building block for replacing MIOpen's fp32 convolutions (models/FlowNetC.py:22-50 and the other networks'
conv blocks).  Not called by the attack path yet."""
from __future__ import annotations

import torch

from . import _lib as L


def split_bf16x3(x: torch.Tensor) -> torch.Tensor:
    """x (float32, any shape) -> [3, *x.shape] bfloat16 with x == p0 + p1 + p2 exactly."""
    L.require_hip(x, "x")
    if x.dtype != torch.float32:
        raise RuntimeError("split_bf16x3: float32 expected")
    planes = torch.empty((3,) + tuple(x.shape), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().ufr_split_bf16x3(L.ptr(x), L.ptr(planes), x.numel(), L.stream()), "split bf16x3")
    return planes


def gemm_split_nt(a_planes: torch.Tensor, b_planes: torch.Tensor, products: int = 6) -> torch.Tensor:
    """C[M,N] = A[M,K] @ B[N,K]^T from `split_bf16x3` planes ([3,M,K] and [3,N,K])."""
    L.require_hip(a_planes, "a_planes")
    L.require_hip(b_planes, "b_planes")
    if a_planes.dtype != torch.bfloat16 or b_planes.dtype != torch.bfloat16 or a_planes.dim() != 3 or b_planes.dim() != 3:
        raise RuntimeError("gemm_split_nt: [3,M,K] / [3,N,K] bfloat16 planes expected")
    _, M, K = a_planes.shape
    _, N, Kb = b_planes.shape
    if K != Kb or a_planes.shape[0] != 3 or b_planes.shape[0] != 3:
        raise RuntimeError("gemm_split_nt: plane shapes do not match")
    c = torch.empty(M, N, dtype=torch.float32, device=a_planes.device)
    L.check(L.lib().ufr_gemm_split_nt(L.ptr(a_planes), L.ptr(b_planes), L.ptr(c), M, N, K, int(products), L.stream()),
            "split gemm")
    return c

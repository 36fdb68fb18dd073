"""Drop-in for the reference's pybind module `resample2d_cuda`
(models/resample2d_package/resample2d_cuda.cc:6-31): the CALLER allocates every output buffer,
functions return 1, fp32 only (resample2d_kernel.cu:221-234).

Contract of `backward` (both kernels behind it, whatever C and UFR_RESAMPLE_LDS): gradInput1 and gradInput2 are OVERWRITTEN with
the gradients -- what the reference computes into the zero-filled buffers its only caller hands over
(models/resample2d_package/resample2d.py:31-43).  The reference kernel itself atomically ADDS onto gradInput1
(resample2d_kernel.cu:118-121); a caller that relied on accumulating into a non-zero buffer must add the result itself
(INTEGRATION.md 3)."""
from __future__ import annotations

import os

import torch

from . import _lib as L


def _check(**tensors):
    for n, t in tensors.items():
        L.require_hip(t, n)
        if t.dtype != torch.float32:
            raise RuntimeError(f"{n} must be float32 (resample2d is fp32 only)")
        if t.dim() != 4:
            raise RuntimeError(f"{n} must be 4-D")


_WORKSPACES = L.LruDict(16)            # (device, stream, B, H, W) -> int32 workspace of the owner-computes adjoint (sampling boxes per tile)


def _workspace(device, B, H, W):
    # per STREAM: launch A writes it and launch B of the same call reads it, in stream order; FlowNet2's small-displacement branch
    # runs its Resample2d on a second stream (flownets/flownet2.py) at the same shape
    key = (device, torch.cuda.current_stream(device).cuda_stream, B, H, W)
    ws = _WORKSPACES.get(key)
    if ws is None:
        nbytes = int(L.lib().ufr_resample2d_backward_workspace_bytes(B, H, W))
        ws = _WORKSPACES[key] = (torch.empty((nbytes + 15) // 16 * 4, dtype=torch.int32, device=device), nbytes)
    return ws


def forward(input1, input2, output, kernel_size, bilinear):
    _check(input1=input1, input2=input2, output=output)
    B, Cc, Hi, Wi = input1.shape
    b2, two, H, W = input2.shape
    if two != 2 or b2 != B or tuple(output.shape) != (B, Cc, H, W):
        raise RuntimeError("resample2d: expected input2 [B,2,H,W] and output [B,C,H,W]")
    with torch.cuda.device(input1.device):
        L.check(L.lib().ufr_resample2d_forward(L.ptr(input1), L.ptr(input2), L.ptr(output), B, Cc, Hi, Wi,
                                               H, W, int(kernel_size), int(bool(bilinear)), L.stream()),
                "resample2d_cuda.forward")
    return 1


def backward(input1, input2, gradOutput, gradInput1, gradInput2, kernel_size, bilinear):
    _check(input1=input1, input2=input2, gradOutput=gradOutput, gradInput1=gradInput1,
           gradInput2=gradInput2)
    B, Cc, Hi, Wi = input1.shape
    _, _, H, W = input2.shape
    if tuple(gradOutput.shape) != (B, Cc, H, W) or gradInput1.shape != input1.shape \
            or gradInput2.shape != input2.shape:
        raise RuntimeError("resample2d backward: gradient buffer shapes do not match the inputs")
    with torch.cuda.device(input1.device):
        if (int(kernel_size) == 1 and (Hi, Wi) == (H, W) and Cc <= 12 and os.environ.get("UFR_RESAMPLE_LDS", "1") == "1"):
            # the adjoint without global atomics: one owner workgroup per tile of gradInput1 (csrc/resample2d_owner.hip);
            # UFR_RESAMPLE_LDS=3 keeps round 2's LDS-privatised scatter, 0 the reference's direct scatter
            # (the workspace is written by launch A and read by launch B of the same call, in stream order: one per shape)
            ws, nbytes = _workspace(input1.device, B, H, W)
            L.check(L.lib().ufr_resample2d_backward_owner(L.ptr(input1), L.ptr(input2), L.ptr(gradOutput), L.ptr(gradInput1),
                                                          L.ptr(gradInput2), L.ptr(ws), nbytes, B, Cc, H, W, L.stream()),
                    "resample2d_cuda.backward")
            return 1
        L.check(L.lib().ufr_resample2d_backward(L.ptr(input1), L.ptr(input2), L.ptr(gradOutput),
                                                L.ptr(gradInput1), L.ptr(gradInput2), B, Cc, Hi, Wi, H, W,
                                                int(kernel_size), int(bool(bilinear)), L.stream()),
                "resample2d_cuda.backward")
    return 1

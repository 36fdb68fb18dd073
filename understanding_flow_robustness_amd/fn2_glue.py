"""Host side of csrc/fn2_glue.hip: the elementwise glue between FlowNet2's sub-networks (models/flownet2_models.py:122-205) as
three autograd Functions on the HIP kernels instead of ~20 torch operators per stage.

    upscale4(flow, mode, scale, divide)          `upsampleK(flow * div_flow)` / `upsampleK(flow / div_flow)`   (:133-136, :147, :160, :176)
    warp_stage(x, flow, div_flow)                cat(x, Resample2d(x[:, 3:], flow), flow / div_flow, ChannelNorm(x[:, :3] - resampled))
                                                 (:138-145, :150-157)
    fusion_input(x, flow_sd, flow_s2)            cat(x[:, :3], flow_sd, flow_s2, |flow_sd|, |flow_s2|, err_sd, err_s2)   (:183-205)

Resample2d itself is `resample2d_cuda` (csrc/warp_norm.hip, csrc/resample2d_owner.hip) in all of them, forward and adjoint.
They serve the native path only (`FlowNet2.forward` with frozen parameters on a HIP device); the torch spelling stays the
module's fallback and the yardstick of tests/test_fn2_glue_gpu.py.
"""
from __future__ import annotations

import torch

from . import _lib as L
from . import resample2d_cuda


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    L.require_hip(t, name, contiguous=False)
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32")
    return t.contiguous()


class _NormalizePair(torch.autograd.Function):
    """x = cat(x1 - mean, x2 - mean) with the subtraction in float64 (flownet2_models.py:93-96, :124-125; FlowNetC.py:73-79) in
    ONE pass (`ufr_normalize_frames`, bit-exact) instead of two casts up, two subtractions, two casts down and a cat; the adjoint
    is the split of the gradient."""

    @staticmethod
    def forward(ctx, x1, x2, mean64):
        x1, x2 = _f32c(x1, "x1"), _f32c(x2, "x2")
        B, C, H, W = x1.shape
        if x2.shape != x1.shape or mean64.dtype != torch.float64 or mean64.numel() != C:
            raise RuntimeError("normalize_pair: two [B,C,H,W] frames and C float64 means expected")
        stack = x1.new_empty(2 * B, C, H, W)
        with torch.cuda.device(x1.device):
            L.check(L.lib().ufr_normalize_frames(L.ptr(x1), L.ptr(x2), L.ptr(stack), B, B, C, H, W, L.ptr(mean64.contiguous()), L.stream()),
                    "normalize frames")
        ctx.C = C
        # [x1 stack | x2 stack] IS cat(x1n, x2n, dim=1) for one pair; more pairs interleave with one cat
        return stack.view(1, 2 * C, H, W) if B == 1 else torch.cat((stack[:B], stack[B:]), dim=1)

    @staticmethod
    def backward(ctx, g):
        return g[:, :ctx.C], g[:, ctx.C:], None


def normalize_pair(x1, x2, mean64):
    return _NormalizePair.apply(x1, x2, mean64)


class _Upscale4(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flow, bilinear, scale, divide):
        flow = _f32c(flow, "flow")
        B, two, h, w = flow.shape
        if two != 2:
            raise RuntimeError("upscale4: a flow [B,2,h,w] expected")
        ctx.opts = (B, h, w, int(bool(bilinear)), float(scale), int(bool(divide)))
        out = flow.new_empty(B, 2, 4 * h, 4 * w)
        with torch.cuda.device(flow.device):
            L.check(L.lib().ufr_flow_upscale4_forward(L.ptr(flow), L.ptr(out), *ctx.opts, L.stream()), "flow upscale x4")
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        B, h, w = ctx.opts[:3]
        gf = g.new_empty(B, 2, h, w)
        with torch.cuda.device(g.device):
            L.check(L.lib().ufr_flow_upscale4_backward(L.ptr(g), L.ptr(gf), *ctx.opts, L.stream()), "flow upscale x4 backward")
        return gf, None, None, None


def upscale4(flow, bilinear: bool, scale: float, divide: bool = False):
    """x4 upsampling of `flow * scale` (or `flow / scale`): bilinear (align_corners=False) or nearest."""
    return _Upscale4.apply(flow, bilinear, scale, divide)


class _WarpStage(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow, div_flow):
        x, flow = _f32c(x, "x"), _f32c(flow, "flow")
        B, six, H, W = x.shape
        if six != 6 or tuple(flow.shape) != (B, 2, H, W):
            raise RuntimeError("warp_stage: x [B,6,H,W] and flow [B,2,H,W] expected")
        frame2 = x[:, 3:].contiguous()                       # (a view for one pair; Resample2d's kernels take dense [B,3,H,W])
        res = x.new_empty(B, 3, H, W)
        out = x.new_empty(B, 12, H, W)
        with torch.cuda.device(x.device):
            resample2d_cuda.forward(frame2, flow, res, 1, True)
            L.check(L.lib().ufr_fn2_stage_pack(L.ptr(x), L.ptr(res), L.ptr(flow), L.ptr(out), B, H, W, float(div_flow), L.stream()),
                    "FlowNet2 stage pack")
        ctx.save_for_backward(frame2, flow, out)
        ctx.div = float(div_flow)
        return out

    @staticmethod
    def backward(ctx, g):
        frame2, flow, packed = ctx.saved_tensors
        g = g.contiguous()
        B, _, H, W = packed.shape
        gx, gres = g.new_empty(B, 6, H, W), g.new_empty(B, 3, H, W)
        gimg, gflow_rs, gflow = torch.empty_like(frame2), torch.empty_like(flow), torch.empty_like(flow)
        with torch.cuda.device(g.device):
            lib = L.lib()
            L.check(lib.ufr_fn2_stage_unpack_grad(L.ptr(g), L.ptr(packed), L.ptr(gx), L.ptr(gres), B, H, W, L.stream()), "FlowNet2 stage unpack")
            resample2d_cuda.backward(frame2, flow, gres, gimg, gflow_rs, 1, True)
            L.check(lib.ufr_fn2_stage_finish_grad(L.ptr(g), L.ptr(gimg), L.ptr(gflow_rs), L.ptr(gx), L.ptr(gflow), B, H, W, ctx.div,
                                                  L.stream()), "FlowNet2 stage finish")
        return gx, gflow, None


def warp_stage(x, flow, div_flow: float):
    """FlowNet2._warp_stage (flownet2_models.py:138-145) as one Function: [B,12,H,W]."""
    return _WarpStage.apply(x, flow, div_flow)


class _FusionInput(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, flow_sd, flow_s2):
        x, flow_sd, flow_s2 = _f32c(x, "x"), _f32c(flow_sd, "flow_sd"), _f32c(flow_s2, "flow_s2")
        B, six, H, W = x.shape
        if six != 6 or tuple(flow_sd.shape) != (B, 2, H, W) or tuple(flow_s2.shape) != (B, 2, H, W):
            raise RuntimeError("fusion_input: x [B,6,H,W] and two flows [B,2,H,W] expected")
        frame2 = x[:, 3:].contiguous()
        res_sd, res_s2 = x.new_empty(B, 3, H, W), x.new_empty(B, 3, H, W)
        out = x.new_empty(B, 11, H, W)
        with torch.cuda.device(x.device):
            resample2d_cuda.forward(frame2, flow_sd, res_sd, 1, True)
            resample2d_cuda.forward(frame2, flow_s2, res_s2, 1, True)
            L.check(L.lib().ufr_fn2_fusion_pack(L.ptr(x), L.ptr(flow_sd), L.ptr(flow_s2), L.ptr(res_sd), L.ptr(res_s2), L.ptr(out), B, H, W,
                                                L.stream()), "FlowNet2 fusion pack")
        ctx.save_for_backward(frame2, flow_sd, flow_s2, res_sd, res_s2, out)
        return out

    @staticmethod
    def backward(ctx, g):
        frame2, flow_sd, flow_s2, res_sd, res_s2, packed = ctx.saved_tensors
        g = g.contiguous()
        B, _, H, W = packed.shape
        gx = g.new_empty(B, 6, H, W)
        gres_sd, gres_s2 = torch.empty_like(res_sd), torch.empty_like(res_s2)
        gf_sd, gf_s2 = torch.empty_like(flow_sd), torch.empty_like(flow_s2)
        gimg_sd, gimg_s2 = torch.empty_like(frame2), torch.empty_like(frame2)
        grs_sd, grs_s2 = torch.empty_like(flow_sd), torch.empty_like(flow_s2)
        with torch.cuda.device(g.device):
            lib = L.lib()
            L.check(lib.ufr_fn2_fusion_unpack_grad(L.ptr(g), L.ptr(packed), L.ptr(res_sd), L.ptr(res_s2), L.ptr(gx), L.ptr(gres_sd),
                                                   L.ptr(gres_s2), L.ptr(gf_sd), L.ptr(gf_s2), B, H, W, L.stream()), "FlowNet2 fusion unpack")
            resample2d_cuda.backward(frame2, flow_sd, gres_sd, gimg_sd, grs_sd, 1, True)
            # (the owner-computes adjoint keeps ONE workspace per shape: the two calls are ordered on this stream)
            resample2d_cuda.backward(frame2, flow_s2, gres_s2, gimg_s2, grs_s2, 1, True)
            L.check(lib.ufr_fn2_fusion_finish_grad(L.ptr(gimg_sd), L.ptr(gimg_s2), L.ptr(grs_sd), L.ptr(grs_s2), L.ptr(gx), L.ptr(gf_sd),
                                                   L.ptr(gf_s2), B, H, W, L.stream()), "FlowNet2 fusion finish")
        return gx, gf_sd, gf_s2


def fusion_input(x, flow_sd, flow_s2):
    """FlowNetFusion's 11-channel input (flownet2_models.py:183-205) as one Function."""
    return _FusionInput.apply(x, flow_sd, flow_s2)

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


def chunk_major(planes: torch.Tensor) -> torch.Tensor:
    """[3, rows, K] row-major planes -> the same shape holding the [3, K/32, rows, 32] chunk-major image: every
    (128-row, 32-wide) tile becomes one contiguous 8 KB run, i.e. whole 128-byte lines for the tile loads."""
    p, rows, k = planes.shape
    return planes.view(p, rows, k // 32, 32).permute(0, 2, 1, 3).contiguous().view(p, rows, k)


def gemm_split_nt(a_planes: torch.Tensor, b_planes: torch.Tensor, products: int = 6, chunked: bool = False) -> torch.Tensor:
    """C[M,N] = A[M,K] @ B[N,K]^T from `split_bf16x3` planes ([3,M,K] and [3,N,K]; `chunked`: both passed
    through `chunk_major`)."""
    L.require_hip(a_planes, "a_planes")
    L.require_hip(b_planes, "b_planes")
    if a_planes.dtype != torch.bfloat16 or b_planes.dtype != torch.bfloat16 or a_planes.dim() != 3 or b_planes.dim() != 3:
        raise RuntimeError("gemm_split_nt: [3,M,K] / [3,N,K] bfloat16 planes expected")
    _, M, K = a_planes.shape
    _, N, Kb = b_planes.shape
    if K != Kb or a_planes.shape[0] != 3 or b_planes.shape[0] != 3:
        raise RuntimeError("gemm_split_nt: plane shapes do not match")
    c = torch.empty(M, N, dtype=torch.float32, device=a_planes.device)
    L.check(L.lib().ufr_gemm_split_nt(L.ptr(a_planes), L.ptr(b_planes), L.ptr(c), M, N, K, int(products), int(chunked),
                                      L.stream()), "split gemm")
    return c


def _pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def nchw_to_nhwc_split3(x: torch.Tensor) -> torch.Tensor:
    """x [B,C,H,W] float32 -> [3, B*H*W, Cpad] bfloat16 planes (channels zero-padded to a multiple of 32)."""
    L.require_hip(x, "x")
    B, C, H, W = x.shape
    planes = torch.empty(3, B * H * W, _pad32(C), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().ufr_nchw_to_nhwc_split3(L.ptr(x), L.ptr(planes), B, C, H, W, _pad32(C), L.stream()), "nchw -> nhwc split")
    return planes


def conv3x3_weight_planes(weight: torch.Tensor, data_gradient: bool = False) -> torch.Tensor:
    """Conv2d(C, N, 3, 1, 1).weight [N,C,3,3] -> [3, Npad, 9*Cpad] planes in (tap, channel) order; done once, the
    weights are frozen during an attack.  `data_gradient`: the planes of the adjoint convolution (taps flipped,
    channel roles swapped), so the same kernel computes d loss / d input from d loss / d output."""
    if data_gradient:
        weight = weight.flip(2, 3).transpose(0, 1)
    N, C = weight.shape[:2]
    npad, cpad = (N + 127) // 128 * 128, _pad32(C)
    w = torch.zeros(npad, 3, 3, cpad, dtype=torch.float32, device=weight.device)
    w[:N, :, :, :C] = weight.detach().float().permute(0, 2, 3, 1)
    return split_bf16x3(w.reshape(npad, 9 * cpad).contiguous())


def conv3x3_split(x_planes: torch.Tensor, w_planes: torch.Tensor, B: int, H: int, W: int, products: int = 6,
                  chunked: bool = False, wide: bool = False) -> torch.Tensor:
    """-> y [B*H*W, Npad] float32 (NHWC rows).  `chunked`: both plane sets passed through `chunk_major`.
    `wide`: the experimental 128x256-tile kernel (csrc/split_conv_wide.hip; Npad a multiple of 256)."""
    L.require_hip(x_planes, "x_planes")
    L.require_hip(w_planes, "w_planes")
    _, M, cpad = x_planes.shape
    _, npad, k = w_planes.shape
    if M != B * H * W or k != 9 * cpad or x_planes.dtype != torch.bfloat16 or w_planes.dtype != torch.bfloat16:
        raise RuntimeError("conv3x3_split: plane shapes do not match")
    y = torch.empty(M, npad, dtype=torch.float32, device=x_planes.device)
    entry = L.lib().ufr_conv3x3_split_wide if wide else L.lib().ufr_conv3x3_split
    L.check(entry(L.ptr(x_planes), L.ptr(w_planes), L.ptr(y), B, H, W, cpad, npad, int(products), int(chunked), L.stream()),
            "split conv")
    return y


# ------------------------------------------------------------------------------------------------------------------
# Opt-in wiring (UFR_SPLIT_CONV=6|3|1, default off): Conv2d(C, N, 3, 1, 1) with frozen weights through the kernels
# above.  A thin composition of the GPU-tested pieces (planes pass, convolution, its adjoint through the flipped
# weights) plus torch layout copies; the composition itself is checked on the CPU with emulated kernels
# (tests/test_split_conv_wiring_cpu.py).  Not measured inside a network yet: DESIGN.md 10, plan for round 2.
_WEIGHT_PLANES: dict = {}


def split_conv_products() -> int:
    import os
    v = int(os.environ.get("UFR_SPLIT_CONV", "0") or 0)
    if v not in (0, 1, 3, 6):
        raise ValueError("UFR_SPLIT_CONV must be 0 (off), 6, 3 or 1 bf16 products")
    return v


def split_conv_applicable(x: torch.Tensor, conv: torch.nn.Conv2d) -> bool:
    """3x3 / stride 1 / padding 1 layers big enough for 128-row tiles to fill the chip."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv.kernel_size == (3, 3) and conv.stride == (1, 1)
            and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1 and conv.in_channels >= 32
            and conv.out_channels >= 64 and x.shape[0] * x.shape[2] * x.shape[3] >= 4096)


def _weight_planes(weight: torch.Tensor, adjoint: bool) -> torch.Tensor:
    """Pre-split planes of a frozen weight, cached per tensor OBJECT (weak reference + version counter): an address
    reused by the allocator for another tensor can never return stale planes."""
    import weakref
    key = (id(weight), adjoint)
    hit = _WEIGHT_PLANES.get(key)
    if hit is not None and hit[0]() is weight and hit[1] == weight._version:
        return hit[2]
    if len(_WEIGHT_PLANES) >= 512:
        for k in [k for k, v in _WEIGHT_PLANES.items() if v[0]() is None]:
            del _WEIGHT_PLANES[k]
    planes = chunk_major(conv3x3_weight_planes(weight, data_gradient=adjoint))
    _WEIGHT_PLANES[key] = (weakref.ref(weight), weight._version, planes)
    return planes


def _experimental() -> bool:
    """UFR_EXPERIMENTAL=1: the kernels of csrc/split_conv_wide.hip (written after round 1's GPU budget, not yet run)."""
    import os
    return os.environ.get("UFR_EXPERIMENTAL") == "1"


def nchw_to_planes_cm(x: torch.Tensor) -> torch.Tensor:
    """EXPERIMENTAL: == chunk_major(nchw_to_nhwc_split3(x)) in one pass."""
    L.require_hip(x, "x")
    B, C, H, W = x.shape
    planes = torch.empty(3, B * H * W, _pad32(C), dtype=torch.bfloat16, device=x.device)
    L.check(L.lib().ufr_nchw_to_planes_cm(L.ptr(x), L.ptr(planes), B, C, H, W, _pad32(C), L.stream()), "nchw -> planes")
    return planes


def rows_to_nchw(y: torch.Tensor, B: int, N: int, H: int, W: int, bias: torch.Tensor | None = None,
                 slope: float = 1.0) -> torch.Tensor:
    """EXPERIMENTAL: the convolution's rows [B*H*W, Npad] -> [B,N,H,W], optionally + bias and LeakyReLU."""
    L.require_hip(y, "y")
    out = torch.empty(B, N, H, W, dtype=torch.float32, device=y.device)
    L.check(L.lib().ufr_rows_to_nchw(L.ptr(y), L.ptr(bias) if bias is not None else None, L.ptr(out), B, N, H, W,
                                     y.shape[1], float(slope), L.stream()), "rows -> nchw")
    return out


def conv_weight_planes(weight: torch.Tensor) -> torch.Tensor:
    """EXPERIMENTAL companion of `conv_split_general`: Conv2d weight [N,C,KH,KW] -> [3, Npad, KH*KW*Cpad] planes in
    (tap, channel) order."""
    N, C, KH, KW = weight.shape
    npad, cpad = (N + 127) // 128 * 128, _pad32(C)
    w = torch.zeros(npad, KH, KW, cpad, dtype=torch.float32, device=weight.device)
    w[:N, :, :, :C] = weight.detach().float().permute(0, 2, 3, 1)
    return split_bf16x3(w.reshape(npad, KH * KW * cpad).contiguous())


def conv_split_general(x_planes: torch.Tensor, w_planes: torch.Tensor, B: int, Hi: int, Wi: int, kernel: tuple, stride: int,
                       padding: int, products: int = 6, chunked: bool = False) -> torch.Tensor:
    """EXPERIMENTAL: forward convolution with any kernel size / stride -> rows [B*Ho*Wo, Npad] float32."""
    L.require_hip(x_planes, "x_planes")
    L.require_hip(w_planes, "w_planes")
    KH, KW = kernel
    _, Min, cpad = x_planes.shape
    _, npad, k = w_planes.shape
    if Min != B * Hi * Wi or k != KH * KW * cpad:
        raise RuntimeError("conv_split_general: plane shapes do not match")
    Ho, Wo = (Hi + 2 * padding - KH) // stride + 1, (Wi + 2 * padding - KW) // stride + 1
    y = torch.empty(B * Ho * Wo, npad, dtype=torch.float32, device=x_planes.device)
    L.check(L.lib().ufr_conv_split_general(L.ptr(x_planes), L.ptr(w_planes), L.ptr(y), B, Hi, Wi, cpad, npad, KH, KW,
                                           int(stride), int(padding), int(products), int(chunked), L.stream()),
            "split conv (general)")
    return y


def split_conv_ok(x: torch.Tensor, weight: torch.Tensor, stride: int, padding: int) -> bool:
    """The same test for call sites that hold the weight, not the module (the band machinery)."""
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3) and stride == 1
            and padding == 1 and weight.shape[1] >= 32 and weight.shape[0] >= 64 and weight.shape[1] == x.shape[1]
            and x.shape[0] * x.shape[2] * x.shape[3] >= 4096)


def split_conv3x3_forward(x: torch.Tensor, weight: torch.Tensor, products: int) -> torch.Tensor:
    """conv2d(x, weight, padding=1), no bias, NCHW in and out."""
    B, _, H, W = x.shape
    if _experimental():
        wp = _weight_planes(weight, False)
        y = conv3x3_split(nchw_to_planes_cm(x.contiguous()), wp, B, H, W, products, chunked=True, wide=wp.shape[1] % 256 == 0)
        return rows_to_nchw(y, B, weight.shape[0], H, W)
    xp = chunk_major(nchw_to_nhwc_split3(x.contiguous()))
    y = conv3x3_split(xp, _weight_planes(weight, False), B, H, W, products, chunked=True)
    return y.view(B, H, W, -1)[..., : weight.shape[0]].permute(0, 3, 1, 2).contiguous()


def split_conv3x3_input_gradient(gy: torch.Tensor, weight: torch.Tensor, products: int) -> torch.Tensor:
    """d/dx of conv2d(x, weight, padding=1) given d/dy: the same kernel with the flipped, transposed weights."""
    B, _, H, W = gy.shape
    if _experimental():
        wp = _weight_planes(weight, True)
        gx = conv3x3_split(nchw_to_planes_cm(gy.contiguous()), wp, B, H, W, products, chunked=True, wide=wp.shape[1] % 256 == 0)
        return rows_to_nchw(gx, B, weight.shape[1], H, W)
    gp = chunk_major(nchw_to_nhwc_split3(gy.contiguous()))
    gx = conv3x3_split(gp, _weight_planes(weight, True), B, H, W, products, chunked=True)
    return gx.view(B, H, W, -1)[..., : weight.shape[1]].permute(0, 3, 1, 2).contiguous()


class SplitConv3x3(torch.autograd.Function):
    """y = conv2d(x, weight, padding=1) without bias; d/dx through the same kernel; no weight gradient (frozen)."""

    @staticmethod
    def forward(ctx, x, weight, products):
        ctx.weight, ctx.products = weight, products
        return split_conv3x3_forward(x, weight, products)

    @staticmethod
    def backward(ctx, gy):
        return split_conv3x3_input_gradient(gy, ctx.weight, ctx.products), None, None


# ------------------------------------------------------------------------------------------------------------------
# Stride-2 transposed convolutions (the decoder's ConvTranspose2d(., ., 4, 2, 1) and the data gradients of the stride-2
# Conv2d layers) as four ordinary gathers: output pixel (2*qy + oy0, 2*qx + ox0) of phase (oy0, ox0) sums the taps
# ky = ry + 2*ty, ry = (oy0 + p) % 2, reading input row qy + cy - ty with cy = (oy0 + p - ry) / 2.  The plan is pure
# index arithmetic (checked on the CPU against F.conv_transpose2d, tests/test_split_conv_wiring_cpu.py); the kernel
# that executes it (csrc/split_conv_wide.hip, ufr_deconv_split) is EXPERIMENTAL like the rest of that file.
def deconv_plan(kernel: int, padding: int):
    """-> list of 4 phases (oy0, ox0, taps) with taps = [(ky, kx, dy, dx)]: out[2qy+oy0, 2qx+ox0] += w[ky,kx] * x[qy+dy, qx+dx].
    Output size is exactly twice the input size (output_padding = 2 + 2*padding - kernel, 0 or 1)."""
    if not 0 <= 2 + 2 * padding - kernel <= 1:
        raise ValueError("deconv_plan: kernel / padding do not give an output of twice the input size")
    phases = []
    for oy0 in (0, 1):
        for ox0 in (0, 1):
            ry, rx = (oy0 + padding) % 2, (ox0 + padding) % 2
            cy, cx = (oy0 + padding - ry) // 2, (ox0 + padding - rx) // 2
            taps = [(ky, kx, cy - (ky - ry) // 2, cx - (kx - rx) // 2)
                    for ky in range(ry, kernel, 2) for kx in range(rx, kernel, 2)]
            phases.append((oy0, ox0, taps))
    return phases


def deconv_weight_planes(weight: torch.Tensor, padding: int):
    """ConvTranspose2d weight [Cin, Cout, K, K] (== a Conv2d weight [N, C, K, K] read as its own adjoint) ->
    (planes [3, total] bf16, per-phase element offsets, Npad, Cpad): for every phase a chunk-major
    [taps*Cpad/32][Npad][32] image, concatenated."""
    cin, cout, K, _ = weight.shape
    npad, cpad = (cout + 127) // 128 * 128, _pad32(cin)
    images, offsets, total = [], [], 0
    for _, _, taps in deconv_plan(K, padding):
        w = torch.zeros(npad, len(taps), cpad, dtype=torch.float32, device=weight.device)
        for t, (ky, kx, _, _) in enumerate(taps):
            w[:cout, t, :cin] = weight.detach().float()[:, :, ky, kx].t()
        img = w.view(npad, len(taps) * cpad // 32, 32).permute(1, 0, 2).contiguous().view(-1)      # chunk-major
        offsets.append(total)
        total += img.numel()
        images.append(img)
    return split_bf16x3(torch.cat(images)), offsets, npad, cpad


def deconv_split(x_planes_cm: torch.Tensor, w_planes: torch.Tensor, offsets, npad: int, B: int, Hi: int, Wi: int, kernel: int,
                 padding: int, products: int = 6) -> torch.Tensor:
    """EXPERIMENTAL: rows [B*2Hi*2Wi, Npad] of the stride-2 transposed convolution; `x_planes_cm` = chunk-major planes
    of the coarse tensor, `w_planes, offsets, npad` from `deconv_weight_planes`."""
    import ctypes as C
    L.require_hip(x_planes_cm, "x_planes_cm")
    L.require_hip(w_planes, "w_planes")
    _, M, cpad = x_planes_cm.shape
    if M != B * Hi * Wi:
        raise RuntimeError("deconv_split: plane shapes do not match")
    host = []
    for (oy0, ox0, taps), off in zip(deconv_plan(kernel, padding), offsets):
        row = [len(taps), oy0, ox0, off]
        for _, _, dy, dx in taps:
            row += [dy, dx]
        host += row + [0] * (36 - len(row))
    plan = (C.c_long * 144)(*host)
    y = torch.empty(B * 4 * Hi * Wi, npad, dtype=torch.float32, device=x_planes_cm.device)
    L.check(L.lib().ufr_deconv_split(L.ptr(x_planes_cm), L.ptr(w_planes), L.ptr(y), B, Hi, Wi, cpad, npad,
                                     w_planes.shape[1], C.cast(plan, C.c_void_p), int(products), L.stream()), "split deconv")
    return y


# ------------------------------------------------------------------------------------------------------------------
# EXPERIMENTAL (UFR_EXPERIMENTAL=1 together with UFR_SPLIT_CONV): every frozen convolution of the conv / deconv blocks
# through csrc/split_conv_wide.hip -- any square kernel, stride 1 or 2, and ConvTranspose2d(., ., K, 2, p) -- so that
# round 2 can time whole networks on the bf16 pipe before fusing anything.  Composition checked with emulated kernels.
_ANY_PLANES: dict = {}


def _cached(key, make):
    hit = _ANY_PLANES.get(key)
    if hit is None:
        if len(_ANY_PLANES) >= 512:
            _ANY_PLANES.clear()
        hit = _ANY_PLANES[key] = make()
    return hit


def _wkey(weight, kind):
    return (weight.data_ptr(), weight._version, tuple(weight.shape), kind)


def split_any_ok(x: torch.Tensor, weight: torch.Tensor, out_pixels: int) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and weight.shape[2] == weight.shape[3]
            and min(weight.shape[0], weight.shape[1]) >= 32 and out_pixels >= 4096)


def split_conv2d_forward(x, weight, stride, padding, products):
    """conv2d(x, weight, stride, padding) without bias, NCHW in / out, through `conv_split_general`."""
    B, _, Hi, Wi = x.shape
    N, _, K, _ = weight.shape
    wp = _cached(_wkey(weight, "conv"), lambda: chunk_major(conv_weight_planes(weight)))
    rows = conv_split_general(nchw_to_planes_cm(x.contiguous()), wp, B, Hi, Wi, (K, K), stride, padding, products, chunked=True)
    return rows_to_nchw(rows, B, N, (Hi + 2 * padding - K) // stride + 1, (Wi + 2 * padding - K) // stride + 1)


def split_conv2d_input_gradient(gy, weight, in_hw, stride, padding, products):
    """d/dx of conv2d(x, weight, stride, padding); None when the shape has no split kernel (caller falls back)."""
    B, _, Ho, Wo = gy.shape
    N, C, K, _ = weight.shape
    if stride == 1 and K - 1 - padding >= 0:
        wp = _cached(_wkey(weight, "adj"), lambda: chunk_major(conv_weight_planes(weight.flip(2, 3).transpose(0, 1))))
        rows = conv_split_general(nchw_to_planes_cm(gy.contiguous()), wp, B, Ho, Wo, (K, K), 1, K - 1 - padding, products,
                                  chunked=True)
        return rows_to_nchw(rows, B, C, Ho + K - 1 - 2 * padding, Wo + K - 1 - 2 * padding)
    if stride == 2 and tuple(in_hw) == (2 * Ho, 2 * Wo) and 0 <= 2 + 2 * padding - K <= 1:
        wp, offsets, npad, _ = _cached(_wkey(weight, ("deconv", padding)), lambda: deconv_weight_planes(weight, padding))
        rows = deconv_split(nchw_to_planes_cm(gy.contiguous()), wp, offsets, npad, B, Ho, Wo, K, padding, products)
        return rows_to_nchw(rows, B, C, 2 * Ho, 2 * Wo)
    return None


class SplitConv2d(torch.autograd.Function):
    """Frozen Conv2d (square kernel, stride 1 or 2) on the split kernels; falls back to ATen for an adjoint shape
    without a split kernel."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, products):
        ctx.weight, ctx.meta = weight, (tuple(x.shape), int(stride), int(padding), int(products))
        return split_conv2d_forward(x, weight, int(stride), int(padding), int(products))

    @staticmethod
    def backward(ctx, gy):
        in_shape, s, p, products = ctx.meta
        gx = split_conv2d_input_gradient(gy, ctx.weight, in_shape[2:], s, p, products)
        if gx is None:
            gx = torch.ops.aten.convolution_backward(gy, gy.new_empty(in_shape), ctx.weight, None, (s, s), (p, p), (1, 1),
                                                     False, (0, 0), 1, (True, False, False))[0]
        return gx, None, None, None, None


class SplitDeconv2x(torch.autograd.Function):
    """Frozen ConvTranspose2d(Cin, Cout, K, 2, p) with an output of twice the input size: forward = four phase GEMMs,
    input gradient = the stride-2 convolution with the same weight tensor."""

    @staticmethod
    def forward(ctx, x, weight, padding, products):
        B, _, H, W = x.shape
        K = weight.shape[2]
        wp, offsets, npad, _ = _cached(_wkey(weight, ("deconv", int(padding))), lambda: deconv_weight_planes(weight, int(padding)))
        rows = deconv_split(nchw_to_planes_cm(x.contiguous()), wp, offsets, npad, B, H, W, K, int(padding), int(products))
        ctx.weight, ctx.meta = weight, (int(padding), int(products))
        return rows_to_nchw(rows, B, weight.shape[1], 2 * H, 2 * W)

    @staticmethod
    def backward(ctx, gy):
        p, products = ctx.meta
        return split_conv2d_forward(gy, ctx.weight, 2, p, products), None, None, None

"""CPU check of `split_gemm.SplitConv3x3` (the opt-in UFR_SPLIT_CONV wiring): the three HIP entry points are
replaced by torch emulations of what they compute (each one is tested against float64 on the GPU in
tests/test_split_gemm_gpu.py), so this pins the composition: plane layouts, chunk-major images, the adjoint's
flipped / transposed weights, channel padding and the NHWC <-> NCHW copies."""
import pytest
import torch
import torch.nn.functional as F

from understanding_flow_robustness_amd import split_gemm as sg


def _split(x):
    parts, r = [], x.float().clone()
    for _ in range(3):
        p = r.bfloat16()
        parts.append(p)
        r = r - p.float()
    return torch.stack(parts)


def _emu_to_planes(x):
    B, C, H, W = x.shape
    cpad = (C + 31) // 32 * 32
    rows = F.pad(x.permute(0, 2, 3, 1).reshape(B * H * W, C), (0, cpad - C))
    return _split(rows)


def _unchunk(planes):
    p, rows, k = planes.shape
    return planes.view(p, k // 32, rows, 32).permute(0, 2, 1, 3).reshape(p, rows, k)


def _emu_conv(x_planes, w_planes, B, H, W, products=6, chunked=False, wide=False):
    if chunked:
        x_planes, w_planes = _unchunk(x_planes), _unchunk(w_planes)
    x = x_planes.float().sum(0)                                    # exact: the pieces add back to the float32
    w = w_planes.float().sum(0)
    cpad, npad = x.shape[1], w.shape[0]
    x = x.view(B, H, W, cpad).permute(0, 3, 1, 2)
    w = w.view(npad, 3, 3, cpad).permute(0, 3, 1, 2)
    return F.conv2d(x, w, padding=1).permute(0, 2, 3, 1).reshape(B * H * W, npad).contiguous()


@pytest.fixture
def emulated(monkeypatch):
    monkeypatch.setattr(sg, "split_bf16x3", _split)
    monkeypatch.setattr(sg, "nchw_to_nhwc_split3", _emu_to_planes)
    monkeypatch.setattr(sg, "conv3x3_split", _emu_conv)
    sg._WEIGHT_PLANES.clear()
    yield
    sg._WEIGHT_PLANES.clear()


def test_chunk_major_round_trip():
    planes = torch.arange(3 * 6 * 64, dtype=torch.float32).view(3, 6, 64).bfloat16()
    cm = sg.chunk_major(planes)
    assert cm.shape == planes.shape and torch.equal(_unchunk(cm), planes)
    assert torch.equal(cm.view(3, 2, 6, 32)[1, 1, 4], planes[1, 4, 32:])       # [plane][chunk][row][32]


@pytest.mark.parametrize("B,C,H,W,N", [(2, 40, 9, 11, 100), (1, 64, 8, 8, 128)])
def test_split_conv_function_equals_conv2d_and_its_adjoint(emulated, B, C, H, W, N):
    g = torch.Generator().manual_seed(N)
    x = torch.randn(B, C, H, W, generator=g, requires_grad=True)
    w = torch.randn(N, C, 3, 3, generator=g) * 0.1
    y = sg.SplitConv3x3.apply(x, w, 6)
    want = F.conv2d(x, w, padding=1)
    assert y.shape == want.shape
    assert float((y - want).detach().abs().max()) <= 2e-5 * float(want.detach().abs().max())
    gy = torch.randn(want.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    (gx_want,) = torch.autograd.grad(want, x, gy)
    assert float((gx - gx_want).abs().max()) <= 2e-5 * float(gx_want.abs().max())
    assert len(sg._WEIGHT_PLANES) == 2                                         # forward + adjoint planes, cached


def test_split_conv_knob(monkeypatch):
    monkeypatch.delenv("UFR_SPLIT_CONV", raising=False)
    assert sg.split_conv_products() == 0
    monkeypatch.setenv("UFR_SPLIT_CONV", "3")
    assert sg.split_conv_products() == 3
    monkeypatch.setenv("UFR_SPLIT_CONV", "4")
    with pytest.raises(ValueError):
        sg.split_conv_products()
    conv = torch.nn.Conv2d(64, 64, 3, 1, 1)
    assert not sg.split_conv_applicable(torch.zeros(1, 64, 96, 96), conv)      # CPU tensors never qualify


@pytest.mark.parametrize("K,p", [(4, 1), (3, 1), (5, 2), (7, 3)])
def test_deconv_plan_reproduces_conv_transpose2d(K, p):
    """The four-phase gather plan executed with dense torch ops (the HIP kernel executes the same plan)."""
    g = torch.Generator().manual_seed(K)
    B, C, N, H, W = 2, 5, 7, 6, 9
    x = torch.randn(B, C, H, W, generator=g, dtype=torch.float64)
    w = torch.randn(C, N, K, K, generator=g, dtype=torch.float64)
    want = F.conv_transpose2d(x, w, stride=2, padding=p, output_padding=2 + 2 * p - K)
    assert want.shape == (B, N, 2 * H, 2 * W)
    got = torch.zeros_like(want)
    seen = set()
    for oy0, ox0, taps in sg.deconv_plan(K, p):
        for ky, kx, dy, dx in taps:
            seen.add((ky, kx))
            shifted = torch.zeros_like(x)                       # shifted[qy, qx] = x[qy + dy, qx + dx], zero outside
            ys, xs = slice(max(0, -dy), min(H, H - dy)), slice(max(0, -dx), min(W, W - dx))
            yd, xd = slice(max(0, dy), min(H, H + dy)), slice(max(0, dx), min(W, W + dx))
            shifted[:, :, ys, xs] = x[:, :, yd, xd]
            got[:, :, oy0::2, ox0::2] += torch.einsum("bchw,cn->bnhw", shifted, w[:, :, ky, kx])
    assert len(seen) == K * K                                   # every tap belongs to exactly one phase
    assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())
    # a stride-2 Conv2d's data gradient is the same plan on the same weight tensor
    xin = torch.randn(B, N, 2 * H, 2 * W, generator=g, dtype=torch.float64, requires_grad=True)
    y = F.conv2d(xin, w, stride=2, padding=p) if K != 4 else None
    if y is not None and y.shape[2:] == (H, W):
        (gx,) = torch.autograd.grad(y, xin, x)
        assert float((gx - want).abs().max()) <= 1e-12 * float(want.abs().max())


@pytest.mark.parametrize("K,p", [(4, 1), (3, 1)])
def test_deconv_weight_image_as_the_kernel_addresses_it(emulated, K, p):
    """`deconv_weight_planes` + the plan, read back with csrc/split_conv_wide.hip::deconv_split_kernel's address
    arithmetic (weights: off[z] + (kt*N + n)*32 + j with kt = tap*KC + kc; activations: chunk-major [KC][M][32])."""
    g = torch.Generator().manual_seed(K + 40)
    B, C, N, H, W = 2, 40, 70, 5, 6
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, N, K, K, generator=g)
    want = F.conv_transpose2d(x.double(), w.double(), stride=2, padding=p, output_padding=2 + 2 * p - K)
    planes, offsets, npad, cpad = sg.deconv_weight_planes(w, p)
    wflat = planes.float().sum(0).double()                                    # [total]
    xcm = sg.chunk_major(_emu_to_planes(x)).float().sum(0).double()           # chunk-major image of [M][cpad]
    M, KC = B * H * W, cpad // 32
    xcm = xcm.view(KC, M, 32)
    out = torch.zeros(B, 2 * H, 2 * W, npad, dtype=torch.float64)
    pix = torch.arange(M)
    qx, qy, qb = pix % W, (pix // W) % H, pix // (W * H)
    for z, ((oy0, ox0, taps), off) in enumerate(zip(sg.deconv_plan(K, p), offsets)):
        acc = torch.zeros(M, npad, dtype=torch.float64)
        for t, (_, _, dy, dx) in enumerate(taps):
            yi, xi = qy + dy, qx + dx
            ok = (yi >= 0) & (yi < H) & (xi >= 0) & (xi < W)
            src = torch.where(ok, qb * H * W + yi * W + xi, torch.zeros_like(pix))
            for kc in range(KC):
                a = xcm[kc][src] * ok[:, None]                                # [M,32]
                kt = t * KC + kc
                bmat = wflat[off + kt * npad * 32: off + (kt + 1) * npad * 32].view(npad, 32)
                acc += a @ bmat.t()
        out[qb, 2 * qy + oy0, 2 * qx + ox0] = acc
    got = out[..., :N].permute(0, 3, 1, 2)
    assert float((got - want).abs().max()) <= 1e-5 * float(want.abs().max())
    assert float(out[..., N:].abs().max()) == 0.0


# ---- the EXPERIMENTAL any-kernel wiring (split_gemm.SplitConv2d / SplitDeconv2x) with emulated kernels ------------
def _emu_general(x_planes, w_planes, B, Hi, Wi, kernel, stride, padding, products=6, chunked=False):
    if chunked:
        x_planes, w_planes = _unchunk(x_planes), _unchunk(w_planes)
    x, w = x_planes.float().sum(0), w_planes.float().sum(0)
    cpad, npad = x.shape[1], w.shape[0]
    x = x.view(B, Hi, Wi, cpad).permute(0, 3, 1, 2)
    w = w.view(npad, kernel[0], kernel[1], cpad).permute(0, 3, 1, 2)
    y = F.conv2d(x, w, stride=stride, padding=padding)
    return y.permute(0, 2, 3, 1).reshape(-1, npad).contiguous()


def _emu_deconv(x_planes_cm, w_planes, offsets, npad, B, Hi, Wi, kernel, padding, products=6):
    """The address arithmetic of deconv_split_kernel (as in test_deconv_weight_image_as_the_kernel_addresses_it)."""
    wflat = w_planes.float().sum(0).double()
    cpad = x_planes_cm.shape[2]
    M, KC = B * Hi * Wi, cpad // 32
    xcm = x_planes_cm.float().sum(0).double().view(KC, M, 32)
    out = torch.zeros(B, 2 * Hi, 2 * Wi, npad, dtype=torch.float64)
    pix = torch.arange(M)
    qx, qy, qb = pix % Wi, (pix // Wi) % Hi, pix // (Wi * Hi)
    for (oy0, ox0, taps), off in zip(sg.deconv_plan(kernel, padding), offsets):
        acc = torch.zeros(M, npad, dtype=torch.float64)
        for t, (_, _, dy, dx) in enumerate(taps):
            yi, xi = qy + dy, qx + dx
            ok = (yi >= 0) & (yi < Hi) & (xi >= 0) & (xi < Wi)
            src = torch.where(ok, qb * Hi * Wi + yi * Wi + xi, torch.zeros_like(pix))
            for kc in range(KC):
                kt = t * KC + kc
                bmat = wflat[off + kt * npad * 32: off + (kt + 1) * npad * 32].view(npad, 32)
                acc += (xcm[kc][src] * ok[:, None]) @ bmat.t()
        out[qb, 2 * qy + oy0, 2 * qx + ox0] = acc
    return out.view(-1, npad).float()


@pytest.fixture
def emulated_any(emulated, monkeypatch):
    monkeypatch.setattr(sg, "nchw_to_planes_cm", lambda x: sg.chunk_major(_emu_to_planes(x)))
    monkeypatch.setattr(sg, "rows_to_nchw",
                        lambda y, B, N, H, W, bias=None, slope=1.0: y.view(B, H, W, -1)[..., :N].permute(0, 3, 1, 2).contiguous())
    monkeypatch.setattr(sg, "conv_split_general", _emu_general)
    monkeypatch.setattr(sg, "deconv_split", _emu_deconv)
    sg._ANY_PLANES.clear()
    yield
    sg._ANY_PLANES.clear()


@pytest.mark.parametrize("K,s,p,H,W", [(3, 1, 1, 8, 10), (3, 2, 1, 8, 10), (5, 2, 2, 8, 12), (7, 2, 3, 12, 8), (1, 1, 0, 6, 7),
                                       (5, 1, 2, 7, 9)])
def test_split_conv2d_function_any_kernel(emulated_any, K, s, p, H, W):
    g = torch.Generator().manual_seed(K * 10 + s)
    x = torch.randn(2, 40, H, W, generator=g, requires_grad=True)
    w = torch.randn(70, 40, K, K, generator=g) * 0.1
    y = sg.SplitConv2d.apply(x, w, s, p, 6)
    want = F.conv2d(x, w, stride=s, padding=p)
    assert y.shape == want.shape
    assert float((y - want).detach().abs().max()) <= 2e-5 * float(want.detach().abs().max())
    gy = torch.randn(want.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    (gx_want,) = torch.autograd.grad(want, x, gy)
    assert float((gx - gx_want).abs().max()) <= 2e-5 * float(gx_want.abs().max())


@pytest.mark.parametrize("K,p", [(4, 1), (3, 1)])
def test_split_deconv_function(emulated_any, K, p):
    g = torch.Generator().manual_seed(K)
    x = torch.randn(2, 40, 5, 7, generator=g, requires_grad=True)
    w = torch.randn(40, 70, K, K, generator=g) * 0.1
    y = sg.SplitDeconv2x.apply(x, w, p, 6)
    want = F.conv_transpose2d(x, w, stride=2, padding=p, output_padding=2 + 2 * p - K)
    assert y.shape == want.shape
    assert float((y - want).detach().abs().max()) <= 2e-5 * float(want.detach().abs().max())
    gy = torch.randn(want.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    (gx_want,) = torch.autograd.grad(want, x, gy)
    assert float((gx - gx_want).abs().max()) <= 2e-5 * float(gx_want.abs().max())

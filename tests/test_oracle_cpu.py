"""CPU suite, part 1: the C oracle (oracle/ufr_oracle.c) is pinned against golden vectors produced by
the reference's own CPU correlation (tests/golden/make_golden.py) and, where the reference library
could be built (oracle/_ref), against that library directly -- bit for bit."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden, t

CORR_CASES = ["corr_check_defaults_f64", "corr_gradcheck_defaults_f64", "corr_flownetc_small_f32",
              "corr_pwc_small_f32", "corr_flownetc_ragged_f32", "corr_pwc_ragged_f32", "corr_rect_f32"]


def expand_params(p):
    """fixture params -> the 12 ints (kH,kW,patchH,patchW,padH,padW,dilH,dilW,dpH,dpW,dH,dW)."""
    p = [int(v) for v in p]
    if len(p) == 6:
        k, patch, stride, pad, dil, dp = p
        return (k, k, patch, patch, pad, pad, dil, dil, dp, dp, stride, stride)
    kH, kW, pH, pW, sH, sW, padH, padW, dH_, dW_, dpH, dpW = p
    return (kH, kW, pH, pW, padH, padW, dH_, dW_, dpH, dpW, sH, sW)


@pytest.mark.parametrize("case", CORR_CASES)
def test_corr_oracle_matches_reference_golden(oracle, case):
    z = load_golden(case)
    prm = expand_params(z["params"])
    a, b, go = t(z["input1"]), t(z["input2"]), t(z["grad_output"])
    out = oracle.corr_forward(a, b, *prm)
    # identical summation order -> exact equality with the reference's CPU result
    assert torch.equal(out, t(z["output"])), case
    g1, g2 = oracle.corr_backward(a, b, go, *prm)
    assert torch.equal(g1, t(z["grad_input1"])), case
    assert torch.equal(g2, t(z["grad_input2"])), case


def test_corr_oracle_full_size_digest(oracle):
    z = load_golden("corr_flownetc_full_digest")
    g = torch.Generator().manual_seed(int(z["seed"]))
    a = torch.randn(1, 256, 48, 160, generator=g)
    b = torch.randn(1, 256, 48, 160, generator=g)
    prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
    out = oracle.corr_forward(a, b, *prm)
    go = torch.randn(out.shape, generator=g)
    g1, g2 = oracle.corr_backward(a, b, go, *prm)
    assert float(out.double().sum()) == float(z["out_sum"])
    assert float(out.double().abs().sum()) == float(z["out_abs"])
    assert np.array_equal(out.flatten()[t(z["out_idx"])].numpy(), z["out_val"])
    assert float(g1.double().sum()) == float(z["g1_sum"]) and float(g2.double().abs().sum()) == float(z["g2_abs"])
    assert np.array_equal(g1.flatten()[t(z["g_idx"])].numpy(), z["g1_val"])
    assert np.array_equal(g2.flatten()[t(z["g_idx"])].numpy(), z["g2_val"])


def test_corr_oracle_equals_reference_library(oracle):
    if oracle.ref_lib() is None:
        pytest.skip("oracle/_ref was not built (no /root/reference here)")
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.float64):
        a = torch.randn(2, 7, 9, 13, dtype=dtype, generator=g)
        b = torch.randn(2, 7, 9, 13, dtype=dtype, generator=g)
        for prm in [(1, 1, 5, 5, 0, 0, 1, 1, 2, 2, 1, 1), (3, 3, 3, 3, 5, 5, 2, 2, 2, 2, 2, 2),
                    (2, 3, 4, 3, 1, 2, 1, 1, 1, 3, 1, 2)]:
            o1, o2 = oracle.corr_forward(a, b, *prm), oracle.corr_forward(a, b, *prm, use_ref=True)
            assert torch.equal(o1, o2), prm
            go = torch.randn(o1.shape, dtype=dtype, generator=g)
            x, y = oracle.corr_backward(a, b, go, *prm), oracle.corr_backward(a, b, go, *prm, use_ref=True)
            assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1]), prm


def test_correlate_wrapper_golden(oracle):
    """models/submodules.py:124-138: view [B,441,H,W] (ph-major) and divide by C."""
    z = load_golden("correlate_wrapper_f32")
    a, b = t(z["input1"]), t(z["input2"])
    out = oracle.spatial_correlation_sample(a, b, kernel_size=1, patch_size=21, stride=1, padding=0,
                                            dilation_patch=2)
    bsz, ph, pw, h, w = out.shape
    assert torch.equal(out.view(bsz, ph * pw, h, w) / a.size(1), t(z["output"]))


def test_oracle_autograd_function_gradcheck(oracle):
    """grad_check.py:44-55 on the oracle's own autograd Function (float64 finite differences)."""
    g = torch.Generator().manual_seed(3)
    a = torch.randn(2, 2, 6, 6, dtype=torch.float64, generator=g, requires_grad=True)
    b = torch.randn(2, 2, 6, 6, dtype=torch.float64, generator=g, requires_grad=True)
    f = lambda x, y: oracle.spatial_correlation_sample(x, y, 3, 3, 2, 1, 2, 2)
    assert torch.autograd.gradcheck(f, (a, b))


def test_lookup_and_altcorr_oracles_match_corrblock_golden(oracle):
    """CorrBlock (models/raft/corr.py:26-106) pins the lookup restatement and, through the
    all-pairs == on-the-fly identity, the alt_cuda_corr restatement."""
    z = load_golden("raft_corrblock_lookup")
    f1, f2, coords = t(z["fmap1"]), t(z["fmap2"]), t(z["coords"])
    B, C, H, W = f1.shape
    corr = torch.matmul(f1.view(B, C, H * W).transpose(1, 2), f2.view(B, C, H * W)) / np.sqrt(C)
    pyr = [corr.view(B * H * W, 1, H, W)]
    for _ in range(3):
        pyr.append(torch.nn.functional.avg_pool2d(pyr[-1], 2, stride=2))
    assert_close(pyr[0], t(z["level0"]), what="pyramid level 0")
    assert_close(pyr[3], t(z["level3"]), what="pyramid level 3")
    out = oracle.corr_lookup(pyr, coords, 4)
    assert_close(out, t(z["output"]), rtol=1e-4, atol_scale=2e-6, what="lookup")
    # alt_corr: pooled feature maps, coords/2^i, NHWC (corr.py:117-137)
    outs = []
    f2l = f2
    for i in range(4):
        c_i = (coords.permute(0, 2, 3, 1) / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
        (o,) = oracle.altcorr_forward(f1.permute(0, 2, 3, 1).contiguous(),
                                      f2l.permute(0, 2, 3, 1).contiguous(), c_i, 4)
        outs.append(o.squeeze(1))
        if i < 3:
            f2l = torch.nn.functional.avg_pool2d(f2l, 2, stride=2)
    alt = torch.stack(outs, dim=1).reshape(B, -1, H, W) / np.sqrt(C)
    assert_close(alt, t(z["output"]), rtol=1e-4, atol_scale=2e-6, what="alt_corr vs CorrBlock")


def test_altcorr_oracle_backward_is_adjoint(oracle):
    g = torch.Generator().manual_seed(9)
    B, H, W, C = 1, 5, 6, 8
    f1 = torch.randn(B, H, W, C, generator=g)
    f2 = torch.randn(B, H, W, C, generator=g)
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    coords = torch.stack([xs, ys], -1) + torch.randn(B, 1, H, W, 2, generator=g)
    (o,) = oracle.altcorr_forward(f1, f2, coords, 2)
    go = torch.randn(o.shape, generator=g)
    g1, g2, gc = oracle.altcorr_backward(f1, f2, coords, go, 2)
    assert float(gc.abs().max()) == 0.0   # never written in the reference (:307,:323)
    # <go, d fwd(f1 + e*d1)> == <g1, d1>
    d1 = torch.randn(f1.shape, generator=g)
    d2 = torch.randn(f2.shape, generator=g)
    (o1,) = oracle.altcorr_forward(f1 + d1, f2, coords, 2)
    (o2,) = oracle.altcorr_forward(f1, f2 + d2, coords, 2)
    assert abs(float(((o1 - o) * go).sum()) - float((g1 * d1).sum())) < 1e-3 * float((g1 * d1).abs().sum())
    assert abs(float(((o2 - o) * go).sum()) - float((g2 * d2).sum())) < 1e-3 * float((g2 * d2).abs().sum())


def test_resample2d_oracle_properties(oracle):
    """No reference test exists (parity unpinned by the reference): check what the source text
    implies -- zero flow is the identity, integer flow is a shift with border clamp, and the
    image-gradient is the adjoint of the forward for fractional parts where trunc == floor."""
    g = torch.Generator().manual_seed(4)
    img = torch.rand(2, 3, 7, 9, generator=g)
    zero = torch.zeros(2, 2, 7, 9)
    out = torch.empty_like(img)
    oracle.resample2d_forward(img, zero, out, 1, True)
    assert torch.equal(out, img)
    flow = torch.zeros(2, 2, 7, 9)
    flow[:, 0] = 2.0
    flow[:, 1] = -1.0
    oracle.resample2d_forward(img, flow, out, 1, True)
    ys = (torch.arange(7) - 1).clamp(0, 6)
    xs = (torch.arange(9) + 2).clamp(0, 8)
    assert torch.equal(out, img[:, :, ys][:, :, :, xs])
    oracle.resample2d_forward(img, flow + 0.4, out, 1, False)   # nearest: floor(x+0.5)
    assert torch.equal(out, img[:, :, (torch.arange(7) - 1 + 0).clamp(0, 6)][:, :, :, xs])
    # adjoint (positive coordinates only: there int() == floor())
    flow = torch.rand(2, 2, 7, 9, generator=g) * 1.5
    img.requires_grad_(True)
    o = oracle.Resample2dFunction.apply(img, flow, 1, True)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    d = torch.randn(img.shape, generator=g)
    o2 = torch.empty_like(o)
    oracle.resample2d_forward((img + d).detach().contiguous(), flow, o2, 1, True)
    lhs = float(((o2 - o.detach()) * go).sum())
    rhs = float((img.grad * d).sum())
    assert abs(lhs - rhs) < 1e-4 * (abs(lhs) + 1)


def test_channelnorm_oracle_matches_closed_form(oracle):
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 3, 5, 7, generator=g, requires_grad=True)
    out = oracle.ChannelNormFunction.apply(x, 2)
    ref = x.detach().pow(2).sum(1, keepdim=True).sqrt()
    assert_close(out, ref, rtol=1e-6, atol_scale=1e-7)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    assert_close(x.grad, go * x.detach() / (ref + 1e-9), rtol=1e-6, atol_scale=1e-7)

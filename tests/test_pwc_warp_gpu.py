"""GPU test of PWC-Net's fused backward warp (csrc/pwc_warp.hip) against the reference's spelling
(models/PWCNet.py:164-204: grid arithmetic + two grid_sample calls + threshold + multiply): forward, the
validity mask, both gradients; flows that leave the frame, ragged sizes, a 1-pixel-wide map."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("owner", ["1", "0"], ids=["owner_computes", "scatter"])
@pytest.mark.parametrize("B,C,H,W,spread", [(2, 5, 24, 40, 3.0), (1, 3, 7, 9, 12.0), (1, 2, 6, 1, 1.0), (2, 32, 48, 80, 0.4), (2, 7, 17, 70, 1.5)])
def test_pwc_warp_matches_torch_spelling(B, C, H, W, spread, owner, monkeypatch):
    from understanding_flow_robustness_amd.flownets import pwcnet
    monkeypatch.setenv("UFR_PWC_WARP_OWNER", owner)          # the adjoint without float atomics (the default) / the scatter
    g = torch.Generator().manual_seed(H * 10 + W)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    flo = (spread * torch.randn(B, 2, H, W, generator=g)).to(DEV)
    x1, f1 = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    x2, f2 = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    want = pwcnet._warp_torch(x1, f1)
    got = pwcnet.warp(x2, f2)
    assert type(got.grad_fn).__name__.startswith("_PwcWarp"), "fused kernel not taken"
    assert torch.equal(got == 0, want == 0) or float(((got == 0) != (want == 0)).float().mean()) < 1e-3   # mask decisions
    # float32 tolerance: the sampling coordinate (up to ~W) is good to 1 ulp (7.6e-6 at 80) whichever way the
    # compiler contracts `(v+1)*W-1`; times the neighbour difference of the features (up to ~4) that is 3e-5.
    assert_close(got, want, rtol=1e-5, atol_scale=3e-5, what="warped features")
    go = torch.randn(want.shape, generator=g).to(DEV)
    gx_w, gf_w = torch.autograd.grad(want, (x1, f1), go)
    gx_g, gf_g = torch.autograd.grad(got, (x2, f2), go)
    assert_close(gx_g, gx_w, rtol=1e-5, atol_scale=3e-5, what="d/d features")
    assert_close(gf_g, gf_w, rtol=1e-4, atol_scale=3e-5, what="d/d flow")


def _compressing_flow(B, H, W, factor):
    """Every pixel samples near the frame's centre: `factor` source pixels per cell and direction."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    fx, fy = (W / 2 - xs) * (1 - 1 / factor), (H / 2 - ys) * (1 - 1 / factor)
    return torch.stack((fx, fy)).expand(B, 2, H, W).contiguous()


@pytest.mark.parametrize("factor", [1.5, 3.0, 8.0, 40.0])
def test_pwc_warp_adjoint_of_a_compressing_flow(factor):
    """Owner-computes keeps 16 (source pixel, weight) slots per cell; a flow that piles more corners on a cell (8x: 64, 40x: the whole
    frame on a few cells) sends the tile down the slow path (LDS float atomics).  Both against float64 autograd of the reference's
    spelling."""
    from understanding_flow_robustness_amd.flownets import pwcnet
    B, C, H, W = 2, 6, 40, 72
    g = torch.Generator().manual_seed(int(factor * 10))
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    flo = (_compressing_flow(B, H, W, factor) + 0.3 * torch.rand(B, 2, H, W, generator=g)).to(DEV)
    go = torch.randn(B, C, H, W, generator=g).to(DEV)
    x32, f32 = x.clone().requires_grad_(True), flo.clone().requires_grad_(True)
    gx, gf = torch.autograd.grad(pwcnet.warp(x32, f32), (x32, f32), go)
    x64, f64 = x.double().requires_grad_(True), flo.double().requires_grad_(True)
    wx, wf = torch.autograd.grad(pwcnet._warp_torch(x64, f64), (x64, f64), go.double())
    # float32 sampling coordinates against float64 ones: weights good to ~1e-5 of a cell; a cell sums up to factor^2 of them
    assert_close(gx, wx.float(), rtol=1e-4, atol_scale=1e-4, what="d/d features")
    assert_close(gf, wf.float(), rtol=1e-3, atol_scale=1e-3, what="d/d flow")


def test_pwc_warp_adjoint_is_bit_reproducible():
    """Every cell adds its contributions in the order of their source pixels: two runs of the owner-computes adjoint agree in
    every bit (the scatter's result depends on the order its atomics retire in)."""
    from understanding_flow_robustness_amd.flownets import pwcnet
    B, C, H, W = 2, 32, 48, 160
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    flo = (2.5 * torch.randn(B, 2, H, W, generator=g)).to(DEV)
    go = torch.randn(B, C, H, W, generator=g).to(DEV)
    runs = []
    for _ in range(3):
        gx, gf = torch.full_like(x, float("nan")), torch.empty_like(flo)       # (the owner writes every element: no zero fill)
        pwcnet.warp_backward(x, flo, go, gx, gf)
        runs.append((gx.clone(), gf.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    assert bool(torch.isfinite(runs[0][0]).all())

"""GPU test of PWC-Net's fused backward warp (csrc/pwc_warp.hip) against the reference's spelling
(models/PWCNet.py:164-204: grid arithmetic + two grid_sample calls + threshold + multiply): forward, the
validity mask, both gradients; flows that leave the frame, ragged sizes, a 1-pixel-wide map."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,C,H,W,spread", [(2, 5, 24, 40, 3.0), (1, 3, 7, 9, 12.0), (1, 2, 6, 1, 1.0), (2, 32, 48, 80, 0.4)])
def test_pwc_warp_matches_torch_spelling(B, C, H, W, spread):
    from understanding_flow_robustness_amd.flownets import pwcnet
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

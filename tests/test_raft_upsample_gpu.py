"""GPU test of RAFT's convex upsampling kernel (csrc/convex_upsample.hip) against the reference's torch
spelling (models/raft/raft.py:111-122): forward and both gradients, ragged widths, batch > 1."""
import pytest
import torch
import torch.nn.functional as F

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _torch_upsample(flow, mask):
    N, _, H, W = flow.shape
    mask = torch.softmax(mask.view(N, 1, 9, 8, 8, H, W), dim=2)
    up = F.unfold(8 * flow, [3, 3], padding=1).view(N, 2, 9, 1, 1, H, W)
    up = torch.sum(mask * up, dim=2).permute(0, 1, 4, 2, 5, 3)
    return up.reshape(N, 2, 8 * H, 8 * W)


@pytest.mark.parametrize("N,H,W", [(1, 48, 160), (2, 5, 37), (3, 1, 1), (1, 7, 64)])
def test_convex_upsample_matches_torch(N, H, W):
    from understanding_flow_robustness_amd.flownets.raft import RAFT
    g = torch.Generator().manual_seed(H * 100 + W)
    flow = (5 * torch.randn(N, 2, H, W, generator=g)).to(DEV)
    mask = (3 * torch.randn(N, 576, H, W, generator=g)).to(DEV)
    f1, m1 = flow.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    f2, m2 = flow.clone().requires_grad_(True), mask.clone().requires_grad_(True)
    want = _torch_upsample(f1, m1)
    got = RAFT.upsample_flow(f2, m2)
    assert type(got.grad_fn).__name__.startswith("_ConvexUpsample"), "fused kernel not taken"
    assert_close(got, want, rtol=1e-5, atol_scale=1e-6, what="upsampled flow")
    gu = torch.randn(want.shape, generator=g).to(DEV)
    gf_w, gm_w = torch.autograd.grad(want, (f1, m1), gu)
    gf_g, gm_g = torch.autograd.grad(got, (f2, m2), gu)
    assert_close(gf_g, gf_w, rtol=1e-5, atol_scale=1e-5, what="d/d flow")
    assert_close(gm_g, gm_w, rtol=1e-5, atol_scale=1e-5, what="d/d mask")

"""GPU suite: fused SepConvGRU gate kernels (csrc/gru.hip) against the reference's torch expressions
(models/raft/update.py:61-73), values and every input gradient."""
import pytest
import torch

from conftest import assert_close

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_gru_gate_kernels_match_torch_expressions():
    from understanding_flow_robustness_amd.flownets.raft import _GruBlend, _GruGates
    g = torch.Generator().manual_seed(12)
    B, Ch, H, W = 2, 128, 12, 20
    mk = lambda *s: (2 * torch.randn(*s, generator=g)).to(DEV).requires_grad_(True)
    zr, h, qp = mk(B, 2 * Ch, H, W), mk(B, Ch, H, W), mk(B, Ch, H, W)
    z, rh = _GruGates.apply(zr, h)
    out = _GruBlend.apply(qp, z, h)
    zr2, h2, qp2 = (t.detach().clone().requires_grad_(True) for t in (zr, h, qp))
    z_ref, r_ref = torch.sigmoid(zr2[:, :Ch]), torch.sigmoid(zr2[:, Ch:])
    out_ref = (1 - z_ref) * h2 + z_ref * torch.tanh(qp2)
    assert_close(z, z_ref, rtol=1e-6, atol_scale=1e-6, what="z")
    assert_close(rh, r_ref * h2, rtol=1e-6, atol_scale=1e-6, what="r*h")
    assert_close(out, out_ref, rtol=1e-5, atol_scale=1e-6, what="h'")
    w1 = torch.randn(out.shape, generator=g).to(DEV)
    w2 = torch.randn(out.shape, generator=g).to(DEV)
    ((out * w1).sum() + (rh * w2).sum()).backward()
    ((out_ref * w1).sum() + ((r_ref * h2) * w2).sum()).backward()
    assert_close(zr.grad, zr2.grad, rtol=1e-4, atol_scale=1e-5, what="d/d zr_pre")
    assert_close(h.grad, h2.grad, rtol=1e-4, atol_scale=1e-5, what="d/d h")
    assert_close(qp.grad, qp2.grad, rtol=1e-4, atol_scale=1e-5, what="d/d q_pre")


def test_sepconvgru_matches_unfused_reference_formula():
    """Whole module: stacked z|r convolution + fused gates == the reference's six convolutions."""
    from understanding_flow_robustness_amd.flownets.raft import SepConvGRU
    torch.manual_seed(3)
    gru = SepConvGRU(128, 256).to(DEV)
    g = torch.Generator().manual_seed(4)
    h = torch.randn(1, 128, 16, 24, generator=g).to(DEV).requires_grad_(True)
    x = torch.randn(1, 256, 16, 24, generator=g).to(DEV).requires_grad_(True)
    out = gru(h, x, {})
    def half(hh, tag):
        hx = torch.cat([hh, x2], 1)
        z = torch.sigmoid(getattr(gru, "convz" + tag)(hx))
        r = torch.sigmoid(getattr(gru, "convr" + tag)(hx))
        q = torch.tanh(getattr(gru, "convq" + tag)(torch.cat([r * hh, x2], 1)))
        return (1 - z) * hh + z * q
    h2, x2 = h.detach().clone().requires_grad_(True), x.detach().clone().requires_grad_(True)
    ref = half(half(h2, "1"), "2")
    assert_close(out, ref, rtol=1e-4, atol_scale=1e-5, what="GRU output")
    w = torch.randn(out.shape, generator=g).to(DEV)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    assert_close(h.grad, h2.grad, rtol=1e-3, atol_scale=1e-4, what="d/d h")
    assert_close(x.grad, x2.grad, rtol=1e-3, atol_scale=1e-4, what="d/d x")

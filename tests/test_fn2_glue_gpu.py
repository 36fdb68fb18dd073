"""GPU suite: FlowNet2's glue between its sub-networks (models/flownet2_models.py:122-205) as fused Functions (fn2_glue.py,
csrc/fn2_glue.hip) against the torch spelling of the same lines -- the module's own fallback path -- forward and adjoint."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


@pytest.mark.parametrize("B,h,w", [(1, 16, 32), (2, 7, 5), (1, 112, 256)])
@pytest.mark.parametrize("bilinear,divide", [(True, False), (False, False), (False, True)])
def test_flow_upscale4_matches_interpolate(B, h, w, bilinear, divide):
    """`upsampleK(flow * div_flow)` / `upsample4(flow / div_flow)` (:133-136, :160, :176): torch's align_corners = False bilinear
    weights resp. nearest, and the gather adjoint against autograd through F.interpolate in float64."""
    from understanding_flow_robustness_amd.fn2_glue import upscale4
    g = torch.Generator().manual_seed(h * w)
    flow = torch.randn(B, 2, h, w, generator=g).to(DEV).requires_grad_(True)
    go = torch.randn(B, 2, 4 * h, 4 * w, generator=g).to(DEV)
    got = upscale4(flow, bilinear, 20.0, divide)
    (gg,) = torch.autograd.grad(got, flow, go)
    f64 = flow.detach().double().requires_grad_(True)
    scaled = f64 / 20.0 if divide else f64 * 20.0
    want = F.interpolate(scaled, scale_factor=4, mode="bilinear", align_corners=False) if bilinear else \
        F.interpolate(scaled, scale_factor=4, mode="nearest")
    (gw,) = torch.autograd.grad(want, f64, go.double())
    assert _rel(got, want) <= 1e-6 and _rel(gg, gw) <= 2e-6
    # and bit for bit the float32 torch forward (same order of operations)
    f32 = flow.detach()
    s32 = f32 / 20.0 if divide else f32 * 20.0
    t32 = F.interpolate(s32, scale_factor=4, mode="bilinear", align_corners=False) if bilinear else F.interpolate(s32, scale_factor=4, mode="nearest")
    assert float((got.detach() - t32).abs().max()) <= 2e-6 * float(t32.abs().max())


def _torch_stage(x, flow, div):
    from understanding_flow_robustness_amd.warp_ops import ChannelNorm, Resample2d
    res = Resample2d()(x[:, 3:], flow)
    return torch.cat((x, res, flow / div, ChannelNorm()(x[:, :3] - res)), dim=1)


@pytest.mark.parametrize("B,H,W", [(1, 64, 128), (2, 40, 56)])
def test_warp_stage_equals_the_torch_spelling(B, H, W):
    """flownet2_models.py:138-145 as one Function: the packed 12 channels and both input gradients against the composition of
    Resample2d, ChannelNorm, sub, div and cat that the module's fallback path runs (same Resample2d kernels on both sides)."""
    from understanding_flow_robustness_amd.fn2_glue import warp_stage
    g = torch.Generator().manual_seed(H + W)
    x = torch.rand(B, 6, H, W, generator=g).to(DEV).requires_grad_(True)
    flow = (torch.randn(B, 2, H, W, generator=g) * 3).to(DEV).requires_grad_(True)
    go = torch.randn(B, 12, H, W, generator=g).to(DEV)
    got = warp_stage(x, flow, 20.0)
    gx, gf = torch.autograd.grad(got, (x, flow), go)
    want = _torch_stage(x, flow, 20.0)
    wx, wf = torch.autograd.grad(want, (x, flow), go)
    assert torch.equal(got, want)                              # same arithmetic, same order
    assert _rel(gx, wx) <= 2e-6 and _rel(gf, wf) <= 2e-6


@pytest.mark.parametrize("B,H,W", [(1, 64, 128), (2, 40, 56)])
def test_fusion_input_equals_the_torch_spelling(B, H, W):
    """FlowNetFusion's input (:183-205): cat(x1, flow_sd, flow_s2, |flow_sd|, |flow_s2|, err_sd, err_s2) and its three gradients."""
    from understanding_flow_robustness_amd.fn2_glue import fusion_input
    from understanding_flow_robustness_amd.warp_ops import ChannelNorm, Resample2d
    g = torch.Generator().manual_seed(H * W)
    x = torch.rand(B, 6, H, W, generator=g).to(DEV).requires_grad_(True)
    fsd = (torch.randn(B, 2, H, W, generator=g) * 0.7).to(DEV).requires_grad_(True)
    fs2 = (torch.randn(B, 2, H, W, generator=g) * 4).to(DEV).requires_grad_(True)
    go = torch.randn(B, 11, H, W, generator=g).to(DEV)
    got = fusion_input(x, fsd, fs2)
    grads = torch.autograd.grad(got, (x, fsd, fs2), go)
    cn, rs = ChannelNorm(), Resample2d()
    want = torch.cat((x[:, :3], fsd, fs2, cn(fsd), cn(fs2), cn(x[:, :3] - rs(x[:, 3:], fsd)), cn(x[:, :3] - rs(x[:, 3:], fs2))), dim=1)
    wants = torch.autograd.grad(want, (x, fsd, fs2), go)
    assert torch.equal(got, want)
    for name, a, b in zip(("d x", "d flow_sd", "d flow_s2"), grads, wants):
        assert _rel(a, b) <= 3e-6, f"{name}: {_rel(a, b):.2e}"


@pytest.mark.parametrize("branch_stream", ["1", "0"])
def test_flownet2_with_the_fused_glue_equals_the_torch_glue(monkeypatch, branch_stream):
    """FlowNet2 on the native path with the fused glue (and FlowNet-SD on the second stream) against the same engines strung
    together by torch operators (UFR_FN2_GLUE=0): flow and both image gradients."""
    from argparse import Namespace
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
    args = Namespace(flownet="FlowNet2")
    net = fetch_model(args, synthetic_seed=3).to(DEV).requires_grad_(False)
    g = torch.Generator().manual_seed(8)
    x1, x2 = torch.rand(1, 3, 128, 192, generator=g).to(DEV), torch.rand(1, 3, 128, 192, generator=g).to(DEV)
    go = torch.randn(1, 2, 128, 192, generator=g).to(DEV)
    res = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("UFR_FN2_GLUE", knob)
        monkeypatch.setenv("UFR_FN2_BRANCH_STREAM", branch_stream)
        a, b = x1.clone().requires_grad_(True), x2.clone().requires_grad_(True)
        flow = predict_flow(net, None, a, b, args)
        ga, gb = torch.autograd.grad(flow, (a, b), go)
        torch.cuda.synchronize()
        res[knob] = (flow.detach(), ga, gb)
    for name, a, b in zip(("flow", "d frame 1", "d frame 2"), res["1"], res["0"]):
        # the gradient runs through four floor() warps: a flow value within rounding of an integer lands in another cell for
        # another order of additions, so a few pixels may move (tests/test_models_gpu.py::test_flownet2_vs_reference_wiring)
        tol = 1e-5 if name == "flow" else 2e-2
        assert _rel(a, b) <= tol, f"{name}: {_rel(a, b):.2e}"


def test_input_gradients_survive_the_next_forward():
    """inside FlowNet2's native path the engines hand aliases of their static buffers to each other
    (`_lib.static_handoff`).  The gradients of the stacked frames `x` -- five consumers -- must NOT be such aliases: whichever
    arrives first may be kept by reference in autograd's accumulation buffer, and the next forward would overwrite it.  A frame
    gradient read AFTER a second forward / backward on other frames equals the copy taken right after the first."""
    from argparse import Namespace

    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
    args = Namespace(flownet="FlowNet2")
    net = fetch_model(args, synthetic_seed=3).to(DEV).requires_grad_(False)
    g = torch.Generator().manual_seed(5)
    H, W = 64, 128
    frames = [torch.rand(1, 3, H, W, generator=g).to(DEV) for _ in range(4)]
    go = torch.randn(1, 2, H, W, generator=g).to(DEV)
    a, b = frames[0].clone().requires_grad_(True), frames[1].clone().requires_grad_(True)
    before = dict(L.VENDOR_FALLBACKS)
    predict_flow(net, None, a, b, args).backward(go)
    assert dict(L.VENDOR_FALLBACKS) == before, "the forward left the native path"
    kept = (a.grad.clone(), b.grad.clone())
    c, d = frames[2].clone().requires_grad_(True), frames[3].clone().requires_grad_(True)
    predict_flow(net, None, c, d, args).backward(go)
    torch.cuda.synchronize()
    assert torch.equal(a.grad, kept[0]) and torch.equal(b.grad, kept[1]), "a frame gradient was a view of an engine buffer"
    assert not torch.equal(c.grad, kept[0])
    # the hand-off state is thread-local and empty outside the composition
    assert not L.static_ok() and not L.static_grads_ok()
    import threading
    seen = []
    with L.static_handoff():
        t = threading.Thread(target=lambda: seen.append((L.static_ok(), L.static_grads_ok())))
        t.start(); t.join()
        assert L.static_ok() and L.static_grads_ok()
        with L.static_handoff(input_grads=False):
            assert L.static_ok() and not L.static_grads_ok()
    assert seen == [(False, False)]

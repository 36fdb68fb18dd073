"""GPU suite: FlowNet2's remaining sub-networks on the native kernels (plane_graph.py) -- FlowNetSD, FlowNetFusion and the
conv1-3 prefixes of FlowNetC / FlowNetS (models/flownet2/FlowNetSD.py:12-126, FlowNetFusion.py:12-71, FlowNetS.py:15-104,
FlowNetC.py:10-131) -- against the torch spelling of the same module with the same weights, both judged against a float64
evaluation: output and input gradient."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / float(b.abs().max())


def _frozen(m):
    for p in m.parameters():
        p.requires_grad_(False)
    return m.eval().to(DEV)


def _realistic(m, seed):
    """Xavier leaves the 2-channel heads tiny; give every bias and weight a scale that makes each branch matter."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in m.parameters():
            if p.dim() == 1:
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)
            else:
                fan = p[0].numel() if p.dim() == 4 else p.numel()
                p.copy_(torch.randn(p.shape, generator=g) * (1.6 / fan) ** 0.5)
    return m


def _run(m, x, gy, monkeypatch, engine):
    monkeypatch.setenv("UFR_ENGINE", "1" if engine else "0")
    xr = x.clone().requires_grad_(True)
    out = m(xr)
    out = out[0] if isinstance(out, tuple) else out
    (gx,) = torch.autograd.grad(out, xr, gy.to(out.dtype))
    return out.detach(), gx


def _three_way(make, cin, B, H, W, monkeypatch, out_hw, seed):
    m = _frozen(_realistic(make(), seed))
    m64 = copy.deepcopy(m).double()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    gy = torch.randn(B, 2, *out_hw, generator=g).to(DEV)
    want, gwant = _run(m64, x.double(), gy.double(), monkeypatch, False)
    tout, tg = _run(m, x, gy, monkeypatch, False)
    nout, ng = _run(m, x, gy, monkeypatch, True)
    assert "_ufr_plane_graphs" in m.__dict__, "the native schedule did not run"
    e_out, e_g = _rel(nout, want), _rel(ng, gwant)
    t_out, t_g = _rel(tout, want), _rel(tg, gwant)
    # A LeakyReLU whose pre-activation sits within float32 rounding of zero takes the other slope than the float64 evaluation; at a
    # coarse level (1/64) that one activation lies in the backward cone of ~2 % of the input gradient, which then differs by O(1e-3) of
    # its scale -- in ANY float32 implementation whose sum lands on that side.  torch / MIOpen always showed it on the B = 2 case
    # (3.7e-3); since round 5 orders a stride-2 launch's taps by parity the igemm's sums round to the same side (3.69e-3 both,
    # gpurun r5_final_a / r5_call31; inside the whole suite, where MIOpen's find step picks other algorithms, torch lands on float64's
    # side and the igemm does not, r5_final_b).  Which side a float32 sum lands on is not a property to test; that the adjoint is RIGHT
    # is: a wrong tap, weight or mask would move most of the gradient by O(1).  Gate: 90 % of the entries to rounding against float64,
    # no entry further than 2e-2 of the scale, and the whole gradient to rounding wherever no such activation exists (the other cases).
    scale = float(gwant.abs().max())
    diff = (ng.double() - gwant).abs().flatten() / scale
    sample = diff[:: max(1, diff.numel() // 4_000_000)]
    q90, beyond = float(torch.quantile(sample, 0.90)), float((diff > 5e-6).float().mean())
    e_nt = _rel(ng, tg)
    print(f"native {e_out:.2e} / {e_g:.2e} (90 % within {q90:.2e}, {beyond:.2e} of the entries beyond 5e-6)   torch {t_out:.2e} / {t_g:.2e}   "
          f"native vs torch gradient {e_nt:.2e}")
    # (round 6, ADVICE r5: the worst entry bounded at ~2x the measured flip, 3.7e-3, instead of 2e-2, and the share of entries beyond
    # rounding at the measured ~2 % of the cone instead of 5 %: a defect confined to a border tap or one tile moves its entries by O(1))
    assert e_out <= 5e-6 and q90 <= 5e-6 and e_g <= 8e-3 and beyond <= 3e-2, \
        f"gradient vs float64: max {e_g:.2e}, 90 % within {q90:.2e}, {beyond:.2e} of the entries beyond 5e-6"
    # second call through the cached schedule: same bits
    nout2, ng2 = _run(m, x, gy, monkeypatch, True)
    assert torch.equal(nout, nout2) and torch.equal(ng, ng2)


@pytest.mark.parametrize("B,H,W", [(1, 64, 128), (2, 128, 192)])
def test_flownetsd_schedule(B, H, W, monkeypatch):
    from understanding_flow_robustness_amd.flownets.flownet2 import FlowNetSD
    _three_way(FlowNetSD, 6, B, H, W, monkeypatch, (H // 4, W // 4), 3)


@pytest.mark.parametrize("B,H,W", [(1, 64, 128), (2, 128, 192)])
def test_flownetfusion_schedule(B, H, W, monkeypatch):
    from understanding_flow_robustness_amd.flownets.flownet2 import FlowNetFusion
    _three_way(FlowNetFusion, 11, B, H, W, monkeypatch, (H, W), 5)


@pytest.mark.parametrize("cin,B,H,W", [(12, 1, 64, 128), (12, 2, 128, 192), (3, 4, 64, 64)])
def test_stem_prefix_schedule(cin, B, H, W):
    """conv1 (7x7 stride 2 as a 16-tap launch over the 2x2-unshuffled frame), conv2, conv3 -> conv2 / conv3 features and the
    gradient of the frame, vs the three torch convolutions in float64."""
    from understanding_flow_robustness_amd.flownets.flownet2 import FlowNetS
    from understanding_flow_robustness_amd.plane_graph import stem_graph as _prefix_graph
    from understanding_flow_robustness_amd.plane_graph import run
    net = _frozen(_realistic(FlowNetS(cin), 7))
    n64 = copy.deepcopy(net).double()
    g = torch.Generator().manual_seed(cin + B)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV)
    g2 = torch.randn(B, 128, H // 4, W // 4, generator=g).to(DEV)
    g3 = torch.randn(B, 256, H // 8, W // 8, generator=g).to(DEV)
    xr = x.clone().requires_grad_(True)
    c2, c3 = run(_prefix_graph(net, B, H, W, cin, DEV), xr)
    (gx,) = torch.autograd.grad([c2, c3], xr, [g2, g3])
    xd = x.double().requires_grad_(True)
    w2 = n64.conv2(n64.conv1(xd))
    w3 = n64.conv3(w2)
    (gwant,) = torch.autograd.grad([w2, w3], xd, [g2.double(), g3.double()])
    assert _rel(c2, w2) <= 2e-6 and _rel(c3, w3) <= 2e-6 and _rel(gx, gwant) <= 5e-6


def test_stale_schedule_backward_raises(monkeypatch):
    from understanding_flow_robustness_amd.flownets.flownet2 import FlowNetSD
    monkeypatch.setenv("UFR_ENGINE", "1")
    m = _frozen(FlowNetSD())
    x = torch.randn(1, 6, 64, 64, device=DEV, requires_grad=True)
    (a,) = m(x)
    m(x.detach().clone().requires_grad_(True))
    with pytest.raises(RuntimeError, match="another forward"):
        a.sum().backward()


@pytest.mark.parametrize("which", ["sd", "fusion"])
def test_first_writers_replace_the_zero_fill(which, monkeypatch):
    """`PlaneGraph._plan_first_writers`: the first adjoint launch to reach a gradient segment writes `=`, the later ones `+=`, and the
    backward no longer starts with a fill of every gradient sum.  Against the same schedule built with the fill (UFR_GRAPH_ZERO_ARENA=1),
    bit for bit, over two passes with different inputs (a segment that nobody overwrote would carry the first pass into the second)."""
    from understanding_flow_robustness_amd.flownets.flownet2 import FlowNetFusion, FlowNetSD
    make, cin, out_div = (FlowNetSD, 6, 4) if which == "sd" else (FlowNetFusion, 11, 1)
    B, H, W = 2, 128, 192
    m_new = _frozen(_realistic(make(), 11))
    m_old = copy.deepcopy(m_new)
    g = torch.Generator().manual_seed(12)
    passes = [(torch.randn(B, cin, H, W, generator=g).to(DEV), torch.randn(B, 2, H // out_div, W // out_div, generator=g).to(DEV)) for _ in range(2)]
    res = {}
    for name, m, knob in (("first writers", m_new, "0"), ("zero fill", m_old, "1")):
        monkeypatch.setenv("UFR_GRAPH_ZERO_ARENA", knob)
        res[name] = [_run(m, x, gy, monkeypatch, True) for x, gy in passes]
        graphs = list(m.__dict__["_ufr_plane_graphs"].values())
        assert graphs and all((gr._zero_list is None) == (knob == "1") for gr in graphs)
        if knob == "0":
            fills = sum(len(gr._zero_list) for gr in graphs)
            elems = sum(t.numel() for gr in graphs for t in gr._zero_list)
            total = sum(gr._zero_arena.numel() for gr in graphs)
            print(f"{which}: {fills} fills of {elems} elements left of an arena of {total}")
            assert elems <= 0.1 * total
    for (o1, g1), (o2, g2) in zip(res["first writers"], res["zero fill"]):
        assert torch.equal(o1, o2) and torch.equal(g1, g2)

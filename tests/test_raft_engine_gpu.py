"""GPU suite: RAFT on the native engines -- the BasicEncoder (raft_encoder_engine.py: igemm convolutions + instance / folded
batch normalisation kernels, models/raft/extractor.py:142-215) and the refinement loop (raft_engine.py, models/raft/raft.py:189-228,
update.py) -- against the torch / MIOpen spelling of the same modules with the same weights, judged against float64."""
import copy
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / float(b.double().abs().max())


@pytest.fixture(scope="module")
def raft():
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    args = Namespace(flownet="RAFT")
    net = fetch_model(args, synthetic_seed=2).to(DEV).eval()
    g = torch.Generator().manual_seed(9)
    for m in net.cnet.modules():                       # non-trivial running statistics / affine: the folding must carry them
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g).to(DEV) * 0.1)
            m.running_var.copy_((torch.rand(m.running_var.shape, generator=g).to(DEV) + 0.5))
            m.weight.data.copy_((torch.rand(m.weight.shape, generator=g).to(DEV) + 0.5))
            m.bias.data.copy_(torch.randn(m.bias.shape, generator=g).to(DEV) * 0.1)
    for p in net.parameters():
        p.requires_grad_(False)
    return net, args


@pytest.mark.parametrize("which,n,H,W", [("fnet", 2, 64, 128), ("cnet", 1, 128, 192), ("fnet", 2, 136, 72)])
def test_encoder_engine_matches_the_torch_encoder(raft, monkeypatch, which, n, H, W):
    """Features and the frame gradient of BasicEncoder: engine vs torch / MIOpen, both against float64 (x3)."""
    net, _ = raft
    enc = getattr(net, which)
    g = torch.Generator().manual_seed(H + n)
    x = (torch.rand(n, 3, H, W, generator=g) * 2 - 1).to(DEV)
    go = torch.randn(n, 256, H // 8, W // 8, generator=g).to(DEV)
    outs = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("UFR_ENGINE", knob)
        xi = x.clone().requires_grad_(True)
        y = enc(xi)
        (gx,) = torch.autograd.grad(y, xi, go)
        outs[knob] = (y.detach(), gx)
    assert enc.__dict__.get("_ufr_encoder_engines"), "the encoder did not run on the engine"
    monkeypatch.setenv("UFR_ENGINE", "0")
    engines = enc.__dict__.pop("_ufr_encoder_engines")
    enc64 = copy.deepcopy(enc).double()
    enc.__dict__["_ufr_encoder_engines"] = engines
    xi = x.double().requires_grad_(True)
    y64 = enc64(xi)
    (g64,) = torch.autograd.grad(y64, xi, go.double())
    (y0, g0), (y1, g1) = outs["0"], outs["1"]
    print(f"{which}: features engine {_rel(y1, y64):.2e}, torch fp32 {_rel(y0, y64):.2e}; gradient engine {_rel(g1, g64):.2e}, "
          f"torch fp32 {_rel(g0, g64):.2e} (vs float64); engine vs torch {_rel(g1, g0):.2e}")
    assert _rel(y1, y64) <= max(3 * _rel(y0, y64), 2e-6)
    assert _rel(g1, g64) <= max(3 * _rel(g0, g64), 2e-5)

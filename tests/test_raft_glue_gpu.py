"""GPU suite: RAFT's glue in front of its encoders and its on-the-fly correlation (raft_glue.py, csrc/raft_glue.hip) against the torch
operators it replaces (models/raft/raft.py:128-129; models/raft/corr.py:97-105, :128-129) -- bit for bit, forward and adjoint."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("B,H,W", [(1, 384, 1280), (2, 40, 52), (3, 7, 12)])
def test_normalize_pair_equals_the_torch_spelling(B, H, W):
    from understanding_flow_robustness_amd.raft_glue import normalize_pair
    g = torch.Generator().manual_seed(H + W)
    a, b = (255 * torch.rand(B, 3, H, W, generator=g)).to(DEV), (300 * torch.rand(B, 3, H, W, generator=g) - 20).to(DEV)
    go = torch.randn(2 * B, 3, H, W, generator=g).to(DEV)
    a1, b1 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    got = normalize_pair(a1, b1)
    assert got is not None and type(got.grad_fn).__name__.startswith("_NormalizePair")
    ga, gb = torch.autograd.grad(got, (a1, b1), go)
    a2, b2 = a.clone().requires_grad_(True), b.clone().requires_grad_(True)
    want = torch.cat(((2 * (a2 / 255.0) - 1.0).contiguous(), (2 * (b2 / 255.0) - 1.0).contiguous()), dim=0)
    wa, wb = torch.autograd.grad(want, (a2, b2), go)
    assert torch.equal(got, want) and torch.equal(ga, wa) and torch.equal(gb, wb)
    # only the first frame asks for a gradient (the attacks perturb both, a public caller may not)
    a3 = a.clone().requires_grad_(True)
    (g3,) = torch.autograd.grad(normalize_pair(a3, b), a3, go)
    assert torch.equal(g3, wa)


def _torch_pyramid(fmap, levels):
    outs, cur = [], fmap
    for l in range(levels):
        if l:
            cur = F.avg_pool2d(cur, 2, stride=2)
        outs.append(cur.permute(0, 2, 3, 1).contiguous())
    return outs


@pytest.mark.parametrize("B,C,H,W,levels", [(1, 256, 48, 160, 4), (2, 128, 24, 40, 4), (1, 300, 50, 163, 4), (2, 7, 9, 21, 3), (1, 16, 13, 11, 2),
                                            (1, 256, 48, 160, 1)])
def test_fmap_pyramid_equals_pooling_and_permutes(B, C, H, W, levels):
    """corr.py:97-105 + :128-129: every level and the gradient of the map, with a gradient on every level and with some levels unused
    (ragged sizes: avg_pool2d drops the odd row / column, the pixels beyond it receive their own level's gradient only)."""
    from understanding_flow_robustness_amd.raft_glue import fmap_pyramid
    g = torch.Generator().manual_seed(C + H)
    f = torch.randn(B, C, H, W, generator=g).to(DEV)
    f1, f2 = f.clone().requires_grad_(True), f.clone().requires_grad_(True)
    got, want = fmap_pyramid(f1, levels), _torch_pyramid(f2, levels)
    assert got is not None and len(got) == levels
    for l, (a, b) in enumerate(zip(got, want)):
        assert a.shape == b.shape and torch.equal(a, b), f"level {l}"
    gos = [torch.randn(w.shape, generator=g).to(DEV) for w in want]
    for used in ([True] * levels, [l % 2 == 0 for l in range(levels)], [l == levels - 1 for l in range(levels)]):
        outs_g = [o for o, u in zip(got, used) if u]
        outs_w = [o for o, u in zip(want, used) if u]
        go = [o for o, u in zip(gos, used) if u]
        (gg,) = torch.autograd.grad(outs_g, f1, go, retain_graph=True)
        (gw,) = torch.autograd.grad(outs_w, f2, go, retain_graph=True)
        assert torch.equal(gg, gw), f"gradient with levels {used}: {float((gg - gw).abs().max()):.3e}"


def test_alternate_corr_block_takes_the_fused_pyramid(monkeypatch):
    """AlternateCorrBlock with the one-launch pyramid against the same block on torch's poolings (UFR_RAFT_GLUE=0): lookup and both
    feature-map gradients bit for bit."""
    from understanding_flow_robustness_amd.flownets.raft_corr import AlternateCorrBlock
    from understanding_flow_robustness_amd.flownets.raft import coords_grid
    g = torch.Generator().manual_seed(3)
    f1, f2 = torch.randn(1, 256, 24, 40, generator=g).to(DEV), torch.randn(1, 256, 24, 40, generator=g).to(DEV)
    coords = coords_grid(1, 24, 40, DEV) + (2 * torch.randn(1, 2, 24, 40, generator=g)).to(DEV)
    res = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("UFR_RAFT_GLUE", knob)
        a, b = f1.clone().requires_grad_(True), f2.clone().requires_grad_(True)
        block = AlternateCorrBlock(a, b, radius=4)
        assert (type(block._f2[1].grad_fn).__name__.startswith("_FmapPyramid")) == (knob == "1")
        out = block(coords)
        go = torch.randn(out.shape, generator=torch.Generator().manual_seed(9)).to(DEV)
        ga, gb = torch.autograd.grad(out, (a, b), go)
        res[knob] = (out.detach(), ga, gb)
        assert len(block.pyramid) == 5 and block.pyramid[4][1].shape[-2:] == (1, 2)      # the reference's attribute, on request
    for name, x, y in zip(("lookup", "d fmap1", "d fmap2"), res["1"], res["0"]):
        assert torch.equal(x, y), f"{name}: {float((x - y).abs().max()):.3e}"

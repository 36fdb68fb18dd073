"""GPU suite: the native FlowNetC head (flownetc_engine.py: igemm convolutions on split planes + chunk-major 2-channel
layers) against the torch / MIOpen spelling of the same module (models/FlowNetC.py:121-197) with the same weights."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def net():
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    n = fetch_model(Namespace(flownet="FlowNetC"), synthetic_seed=0).to(DEV)
    for p in n.parameters():
        p.requires_grad_(False)
    return n


def _rel(a, b):
    a, b = a.detach().double(), b.detach().double()
    return float((a - b).abs().max()) / float(b.abs().max())


@pytest.mark.parametrize("B,H,W", [(2, 64, 128), (1, 128, 192)])
def test_engine_head_forward_and_gradients_match_the_torch_head(net, monkeypatch, B, H, W):
    """Flow and the three feature gradients of the head: engine vs the torch / MIOpen spelling, both judged against a
    float64 evaluation of the same module (the engine may be no further from it than MIOpen's fp32 path, x3)."""
    import copy
    g = torch.Generator().manual_seed(3)
    feats = []
    for shape in ((B, 128, H // 4, W // 4), (B, 256, H // 8, W // 8), (B, 256, H // 8, W // 8)):
        feats.append(torch.randn(*shape, generator=g).mul_(0.5).to(DEV))
    gflow = torch.randn(B, 2, H, W, generator=g).to(DEV)
    outs = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("UFR_ENGINE", knob)
        leaves = [f.clone().requires_grad_(True) for f in feats]
        flow = net.head(*leaves)
        grads = torch.autograd.grad(flow, leaves, gflow)
        outs[knob] = (flow.detach(), grads)
    assert "_ufr_head_engines" in net.__dict__ and len(net.__dict__["_ufr_head_engines"]) >= 1
    monkeypatch.setenv("UFR_ENGINE", "0")
    engines = net.__dict__.pop("_ufr_head_engines")              # ctypes descriptors: not copyable
    net64 = copy.deepcopy(net).double()
    net.__dict__["_ufr_head_engines"] = engines
    leaves = [f.double().requires_grad_(True) for f in feats]
    flow64 = net64.head(*leaves)
    grads64 = torch.autograd.grad(flow64, leaves, gflow.double())
    (f0, g0), (f1, g1) = outs["0"], outs["1"]
    print(f"flow: engine {_rel(f1, flow64):.2e}, torch fp32 {_rel(f0, flow64):.2e} of max |flow| (vs float64)")
    assert _rel(f1, flow64) <= max(3 * _rel(f0, flow64), 1e-5)
    for name, a, b, truth in zip(("d/d conv2a", "d/d conv3a", "d/d conv3b"), g1, g0, grads64):
        e_eng, e_t32 = _rel(a, truth), _rel(b, truth)
        print(f"{name}: engine {e_eng:.2e}, torch fp32 {e_t32:.2e} of max |gradient| (vs float64)")
        # a LeakyReLU whose pre-activation sits within rounding of zero takes the other slope in an fp32 evaluation:
        # isolated entries of a few 1e-4, in either implementation
        assert e_eng <= max(3 * e_t32, 5e-4), f"{name}: engine {e_eng:.2e} vs torch fp32 {e_t32:.2e}"
        frac = float(((a.double() - truth).abs() > 1e-4 * float(truth.abs().max())).float().mean())
        assert frac <= 1e-2, f"{name}: {frac:.2e} of the entries beyond 1e-4"


def test_engine_whole_network_vs_reference_golden(net, monkeypatch):
    """FlowNetC forward + image gradients through the engine against the reference's golden vectors."""
    import numpy as np
    from conftest import assert_close, load_golden, t
    monkeypatch.setenv("UFR_ENGINE", "1")
    z = load_golden("flownetc_fwd_64x128")
    x1, x2 = t(z["x1"], DEV).requires_grad_(True), t(z["x2"], DEV).requires_grad_(True)
    flow = net(x1, x2)
    assert_close(flow, t(z["flow"]), rtol=1e-4, atol_scale=1e-4, what="flow")
    loss = (1 - torch.nn.functional.cosine_similarity(flow, t(z["target"], DEV))).mean()
    g1, g2 = torch.autograd.grad(loss, (x1, x2))
    assert_close(g1, t(z["g1"]), rtol=1e-3, atol_scale=2e-4, what="d loss / d frame 1")
    assert_close(g2, t(z["g2"]), rtol=1e-3, atol_scale=2e-4, what="d loss / d frame 2")


def test_engine_attack_matches_reference_trace(net, monkeypatch):
    """The 2-iteration attack() golden (patch_attacks/main.py:523-613 run on the reference) with the engine on."""
    from conftest import load_golden, t
    from understanding_flow_robustness_amd.patch_attack import attack
    monkeypatch.setenv("UFR_ENGINE", "1")
    from test_flow_oracle_cpu import ATTACK_CASES
    z = load_golden("attack_flownetc_64x128")
    for name, l2, lr in ATTACK_CASES:
        args = Namespace(flownet="FlowNetC", l2=l2, alpha=0.0, lr=lr, max_count=2)
        patch = t(z["patch0"], DEV).clone()
        attack(net, t(z["tgt"], DEV), None, t(z["ref"], DEV), patch, t(z["mask"], DEV), t(z["patch0"], DEV),
               t(z["target"], DEV), None, args=args, use_graph=(name == "cos_lr1000"))
        ref_patch = t(z[f"{name}_it2_patch"])
        upd = float((ref_patch - t(z["patch0"])).abs().max())
        err = float((patch.cpu() - ref_patch).abs().max())
        assert err <= 1e-4 * max(upd, 1.0), f"{name}: patch err {err:.3e}, update {upd:.3e}"


def test_engine_banded_step_equals_full_frame_torch_step(net, monkeypatch):
    """The whole windowed attack step (prefix window, column band, incremental head forward, windowed correlation adjoint)
    at the benchmark frame size with the engine on, against the full-frame torch step: 4 pairs behind one patch."""
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    B, H, W = 4, 384, 1280
    g = torch.Generator().manual_seed(11)
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    target = torch.randn(B, 2, H, W, generator=g).to(DEV)
    yy, xx = torch.meshgrid(torch.arange(51), torch.arange(51), indexing="ij")
    mask_p = (((yy - 25) ** 2 + (xx - 25) ** 2) <= 23 ** 2).float().expand(1, 3, 51, 51).contiguous().to(DEV)
    patch0 = torch.rand(1, 3, 51, 51, generator=g).to(DEV)
    placements = ([(0, 0), (333, 1229), (0, 600), (170, 640)], [(333, 0), (160, 640), (7, 1221), (100, 300)])

    def run(engine, cone, lr, iters, graph):
        monkeypatch.setenv("UFR_ENGINE", "1" if engine else "0")
        args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=lr, max_count=iters)
        step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51), use_cone=cone, use_graph=graph)
        outs = []
        for origins in placements:
            step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
            n, loss = step.run(iters)
            outs.append((step.patch.clone(), n, loss))
        return step, outs

    _, probe = run(False, False, 1.0, 1, False)
    lr = 0.5 / float((probe[0][0] - patch0).abs().max())          # first update peaks at 0.5: the +-2 clamp stays inactive
    _, full = run(False, False, lr, 3, False)
    step, eng = run(True, True, lr, 3, True)
    assert step.cone is not None and step.band is not None and step.band.width == 608 and step.graph_next is not None
    assert step.eng is not None                        # cached features resident in the native head's planes
    for (pf, nf, lf), (pe, ne, le) in zip(full, eng):
        upd = float((pf - patch0).abs().max())
        err = (pf - pe).abs()
        off = float((err > 1e-4 * upd).float().mean())
        print(f"engine windowed step vs full-frame torch step: worst {float(err.max()) / upd:.2e} of the update, {off:.2%} beyond 1e-4")
        assert nf == ne and abs(lf - le) <= 1e-4 * max(abs(lf), 1.0)
        assert off <= 0.05 and float(err.max()) <= 5e-3 * upd


@pytest.mark.parametrize("B,C,H,W", [(2, 194, 96, 320), (16, 70, 20, 100), (1, 386, 48, 160), (2, 1026, 12, 40), (3, 37, 7, 9),
                                     (8, 194, 96, 320), (8, 386, 48, 160)])
def test_flow_head_planes_kernels_vs_torch(B, C, H, W):
    """csrc/engine_small.hip: predict_flow (Conv2d(C,2,3,1,1)) on the chunk-major planes -- the LDS-tiled kernel of the big
    grids, the per-pixel kernel of the small ones, ragged tiles -- and its data gradient (write and accumulate), against
    torch in float64."""
    import torch.nn.functional as F
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    from understanding_flow_robustness_amd.flownetc_engine import _pack_flow_head, _pack_flow_head_mfma
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    w = (torch.randn(2, C, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.randn(2, generator=g).to(DEV)
    chunks = ig.pad32(C) // 32
    pl = ig.Planes(B, H, W, chunks + 1, DEV).load_nchw(x, chunk0=1)
    wpk = _pack_flow_head(w)
    out = torch.empty(B, 2, H, W, device=DEV)
    L.check(L.lib().ufr_flow_head_planes_forward(L.ptr(pl.t), pl.plane_stride, 1, chunks, L.ptr(wpk), chunks, L.ptr(b), L.ptr(out), B, H, W,
                                                 L.stream()))
    want = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
    assert _rel(out, want) <= 1e-5, f"forward {_rel(out, want):.2e}"
    # the matrix-core form (per-pixel GEMM + 9-tap gather): tile heights 8 / 4 / 2, one and two chunk slices
    out_m = torch.full((B, 2, H, W), float("nan"), device=DEV)
    L.check(L.lib().ufr_flow_head_planes_forward_mfma(L.ptr(pl.t), pl.plane_stride, 1, chunks, L.ptr(_pack_flow_head_mfma(w)), chunks, L.ptr(b),
                                                      L.ptr(out_m), B, H, W, L.stream()))
    assert _rel(out_m, want) <= 1e-5, f"forward (mfma) {_rel(out_m, want):.2e}"
    gy = torch.randn(B, 2, H, W, generator=g).to(DEV)
    x0 = torch.zeros(B, C, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
    (gx,) = torch.autograd.grad(F.conv2d(x0, w.double(), None, 1, 1), x0, gy.double())
    G = ig.GradSum(B, H, W, chunks + 1, DEV)
    G.t.fill_(0.5)
    L.check(L.lib().ufr_flow_head_planes_backward(L.ptr(gy), L.ptr(wpk), chunks, L.ptr(G.t), G.chunks, 1, chunks, B, H, W, 0, L.stream()))
    assert _rel(G.to_nchw(C, 1), gx) <= 1e-5
    assert bool((G.t[0] == 0.5).all())                                   # the neighbouring chunk is untouched
    L.check(L.lib().ufr_flow_head_planes_backward(L.ptr(gy), L.ptr(wpk), chunks, L.ptr(G.t), G.chunks, 1, chunks, B, H, W, 1, L.stream()))
    # ABI 7: a chunk range that leaves the planes / the packed weights / the gradient sum is refused, nothing is launched
    lib = L.lib()
    for rc in (lib.ufr_flow_head_planes_forward_mfma(L.ptr(pl.t), pl.plane_stride, 2, chunks, L.ptr(_pack_flow_head_mfma(w)), chunks,
                                                     L.ptr(b), L.ptr(out_m), B, H, W, L.stream()),
               lib.ufr_flow_head_planes_forward_mfma(L.ptr(pl.t), pl.plane_stride, 0, chunks + 1, L.ptr(_pack_flow_head_mfma(w)), chunks,
                                                     L.ptr(b), L.ptr(out_m), B, H, W, L.stream()),
               lib.ufr_flow_head_planes_forward(L.ptr(pl.t), pl.plane_stride, 2, chunks, L.ptr(wpk), chunks, L.ptr(b), L.ptr(out), B, H, W,
                                                L.stream()),
               lib.ufr_flow_head_planes_backward(L.ptr(gy), L.ptr(wpk), chunks, L.ptr(G.t), G.chunks, 2, chunks, B, H, W, 0, L.stream())):
        assert rc == -1 and b"chunks" in lib.ufr_last_error()
    assert _rel(G.to_nchw(C, 1), 2 * gx) <= 1e-5


@pytest.mark.parametrize("B,H,W", [(2, 24, 40), (1, 48, 160), (3, 13, 37), (1, 20, 200)])
def test_correlation_on_planes_equals_the_reference_cost_volume(B, H, W):
    """csrc/correlation_planes.hip (banded GEMM on the matrix cores, planes in / planes out, / C and LeakyReLU fused) against
    the cost volume of the spatial correlation sampler (pinned bit for bit to the reference's CPU implementation) in
    float64: row widths of 2, 4 and 5 waves per block, several column blocks, odd sizes."""
    import torch.nn.functional as F
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    from understanding_flow_robustness_amd import spatial_correlation_sampler_backend as be
    g = torch.Generator().manual_seed(9)
    f1, f2 = torch.randn(B, 256, H, W, generator=g).to(DEV), torch.randn(B, 256, H, W, generator=g).to(DEV)
    a = ig.Planes(B, H, W, 8, DEV).load_nchw(f1)
    b = ig.Planes(B, H, W, 8, DEV).load_nchw(f2)
    out = ig.Planes(B, H, W, 15, DEV)
    out.t.fill_(3.0)
    L.check(L.lib().ufr_corr_forward_planes(L.ptr(a.t), L.ptr(b.t), a.plane_stride, L.ptr(out.t), out.plane_stride, 1, B, 256, H, W,
                                            21, 2, 1.0 / 256.0, 0.1, L.stream()))
    want = F.leaky_relu(be.forward(f1.double(), f2.double(), 1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1).view(B, 441, H, W) / 256.0, 0.1)
    got = out.to_nchw(441, 1)
    assert _rel(got, want) <= 1e-5, f"cost volume {_rel(got, want):.2e}"
    assert bool((out.t[:, 0] == 3.0).all())                              # conv_redir's chunk is untouched
    assert bool((out.t[:, 14, :, 25:] == 3.0).all())                     # channels 441..447: never written (the engine keeps them zero)


@pytest.mark.parametrize("B,H,W,wh,ww,margin,origins", [
    (2, 48, 160, 16, 16, 2, [(128, 512), (0, 0)]),            # interior window; window on the frame's corner (no rim there)
    (3, 24, 40, 16, 16, 1, [(64, 192), (8, 8), (40, 100)]),    # clamped origins, window reaching the right / bottom edge
    (1, 20, 36, 10, 12, 0, [(24, 56)]),                        # ragged window (fewer than 16 cells), no rim
])
def test_corr_backward_window_fused_equals_the_unfused_chain(B, H, W, wh, ww, margin, origins):
    """csrc/correlation_window_mfma.hip (both adjoints of the cost volume on the window's cells, on the matrix cores, read
    from the chunk-major gradient sums, + conv_redir's gradient, written window-sized with the rim zeroed) against the
    chain it replaces: gradient sum -> NCHW, the full correlation backward (pinned to the reference's CPU implementation
    in tests/test_ops_gpu.py), + conv_redir's gradient, ufr_window_gather."""
    import ctypes as C
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    from understanding_flow_robustness_amd import spatial_correlation_sampler_backend as correlation
    g = torch.Generator().manual_seed(11)
    f1, f2 = torch.randn(B, 256, H, W, generator=g).to(DEV), torch.randn(B, 256, H, W, generator=g).to(DEV)
    gc = torch.randn(B, 441, H, W, generator=g).to(DEV)                  # d loss / d (cost volume / C ... before the 1/C)
    gr = torch.randn(B, 256, H, W, generator=g).to(DEV)                  # conv_redir's input gradient
    M = B * H * W

    def chunked(x, chunks, chunk0):                                       # NCHW -> float32 [chunks][M][32]
        full = torch.zeros(B, chunks * 32, H, W, device=DEV)
        full[:, chunk0 * 32:chunk0 * 32 + x.shape[1]] = x
        return full.view(B, chunks, 32, H, W).permute(1, 0, 3, 4, 2).reshape(chunks, M, 32).contiguous()
    G, Gr = chunked(gc, 15, 1), chunked(gr, 8, 0)
    win = torch.zeros(B, 8, dtype=torch.int32, device=DEV)
    for n, (oy, ox) in enumerate(origins):
        win[n, 0], win[n, 1] = oy, ox
    got = torch.full((2 * B, 256, wh, ww), float("nan"), device=DEV)
    L.check(L.lib().ufr_corr_backward_window_fused(L.ptr(f1), L.ptr(f2), L.ptr(G), 1, 1.0 / 256.0, L.ptr(Gr), L.ptr(got), B, 256, H, W,
                                                   21, 2, L.ptr(win), 8, wh, ww, margin, L.stream()))
    p = correlation._params(1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
    g1, g2 = torch.empty_like(f1), torch.empty_like(f2)
    gcs = (gc / 256.0).contiguous()
    L.check(L.lib().ufr_corr_backward(L.ptr(f1), L.ptr(f2), L.ptr(gcs), L.ptr(g1), L.ptr(g2), L.UFR_F32, B, 256, H, W, C.byref(p),
                                      L.stream()))
    g1 += gr
    want = torch.empty(2 * B, 256, wh, ww, device=DEV)
    for src, dst in ((g1, want[:B]), (g2, want[B:])):
        L.check(L.lib().ufr_window_gather(L.ptr(src), L.ptr(dst), L.ptr(win), B, B, 256, H, W, wh, ww, 8, margin, L.stream()))
    assert bool(torch.isfinite(got).all())
    assert bool(((want == 0) == (got == 0)).all()) or margin == 0        # the rim is zero in both
    assert _rel(got[:B], want[:B]) <= 1e-5 and _rel(got[B:], want[B:]) <= 1e-5, (_rel(got[:B], want[:B]), _rel(got[B:], want[B:]))


def test_window_prefix_on_the_engine_equals_the_torch_prefix(net):
    """flownetc_engine.py `window_prefix_forward` / `window_prefix_backward` (conv2 / conv3 of the attack's 128 x 128
    window and their data gradients on the igemm, conv1 on torch) against torch autograd through `net.encode` on the
    same window stack, judged against a float64 evaluation."""
    from understanding_flow_robustness_amd.flownetc_engine import get_engine
    B, H, W, wh, ww = 2, 128, 256, 128, 128
    g = torch.Generator().manual_seed(21)
    eng = get_engine(net, B, H, W, DEV)
    xw = torch.rand(2 * B, 3, wh, ww, generator=g).mul_(255.0).to(DEV)
    win = torch.zeros(B, 8, dtype=torch.int32, device=DEV)
    win[:, 1] = 64
    eng.window_prefix_forward(xw, win, 0, 0)
    P = eng._wprefix
    x32 = xw.clone().requires_grad_(True)
    c2, c3 = net.encode(x32)
    import torch.nn.functional as F

    def encode64(x):                                                       # models/FlowNetC.py:100-119 in float64
        y = x - net._mean64.double()
        outs = []
        for name, k in (("conv1", 7), ("conv2", 5), ("conv3", 5)):
            conv = getattr(net, name)[0]
            y = F.leaky_relu(F.conv2d(y, conv.weight.double(), conv.bias.double(), 2, (k - 1) // 2), 0.1)
            outs.append(y)
        return outs[1], outs[2]
    x64 = xw.double().requires_grad_(True)
    c2_64, c3_64 = encode64(x64)
    # conv1 on the igemm over the packed planes (pixel-unshuffle, two columns per chunk, padding inside the buffer)
    c1_64 = F.leaky_relu(F.conv2d(x64 - net._mean64.double(), net.conv1[0].weight.double(), net.conv1[0].bias.double(), 2, 3), 0.1)
    c1_32 = F.leaky_relu(F.conv2d(net.normalize_correctly(xw), net.conv1[0].weight, net.conv1[0].bias, 2, 3), 0.1)
    e_eng, e_t = _rel(P["c1"].to_nchw(64, 0), c1_64.detach()), _rel(c1_32, c1_64.detach())
    print(f"conv1: engine {e_eng:.2e}, torch fp32 {e_t:.2e} (vs float64)")
    assert ("conv1" in P or "direct" in P) and e_eng <= max(3 * e_t, 2e-6)
    for name, got, t32, t64 in (("conv2", P["c2_nchw"], c2, c2_64), ("conv3", P["c3_nchw"], c3, c3_64)):
        e_eng, e_t = _rel(got, t64), _rel(t32.detach(), t64)
        print(f"{name}: engine {e_eng:.2e}, torch fp32 {e_t:.2e} (vs float64)")
        assert e_eng <= max(3 * e_t, 2e-6), name
    gw2 = torch.randn(B, 128, wh // 4, ww // 4, generator=g).to(DEV)
    gw3 = torch.randn(2 * B, 256, wh // 8, ww // 8, generator=g).to(DEV)
    gw2_all = torch.cat((gw2, torch.zeros_like(gw2)), 0)
    gx = eng.window_prefix_backward(gw3, gw2)
    (gx32,) = torch.autograd.grad((c2, c3), x32, (gw2_all, gw3))
    (gx64,) = torch.autograd.grad((c2_64, c3_64), x64, (gw2_all.double(), gw3.double()))
    e_eng, e_t = _rel(gx, gx64), _rel(gx32, gx64)
    print(f"d/d window: engine {e_eng:.2e}, torch fp32 {e_t:.2e} (vs float64)")
    # LeakyReLU slope flips at pre-activations within rounding of zero: isolated entries, in either implementation
    assert e_eng <= max(3 * e_t, 5e-4)
    frac = float(((gx.double() - gx64).abs() > 1e-4 * float(gx64.abs().max())).float().mean())
    assert frac <= 1e-2, frac


@pytest.mark.parametrize("Ba,Bb,H,W", [(2, 0, 64, 128), (1, 2, 120, 120), (3, 1, 48, 200), (8, 0, 384, 1280), (1, 0, 16, 64), (1, 1, 34, 70)])
def test_conv1_direct_kernel_vs_float64(net, Ba, Bb, H, W):
    """csrc/conv1_direct.hip: Conv2d(3, 64, 7, 2, 3) + bias + LeakyReLU of models/FlowNetC.py:100-104 from the RAW frames (mean
    subtraction of normalize_correctly :73-79 and the zero padding inside the kernel) into conv1's planes, against a float64
    evaluation; judged like every engine layer by torch's own float32 error.  Sizes with partial tiles (the 120 x 120 attack
    window, 34 x 70), two frame stacks, the bench's 8 x 384 x 1280; a second chunk offset leaves the neighbours untouched."""
    import torch.nn.functional as F

    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    g = torch.Generator().manual_seed(5 + H)
    fa = torch.rand(Ba, 3, H, W, generator=g).to(DEV)
    fb = torch.rand(Bb, 3, H, W, generator=g).to(DEV) if Bb else None
    conv = net.conv1[0]
    wimg = ig.conv1_direct_weights(conv.weight)
    bias = conv.bias.detach().float().contiguous()
    mean = net._mean64.reshape(-1).contiguous()
    n = Ba + Bb
    planes = ig.Planes(n, H // 2, W // 2, 4, DEV)                      # conv1 at chunks 1-2 of a wider buffer
    planes.t.fill_(7.0)
    L.check(L.lib().ufr_conv1_direct(L.ptr(fa), L.ptr(fb) if Bb else None, Ba, Bb, H, W, L.ptr(mean), L.ptr(wimg), L.ptr(bias), 0.1,
                                     L.ptr(planes.t), planes.plane_stride, 1, L.stream()))
    x = fa if not Bb else torch.cat((fa, fb))
    want64 = F.leaky_relu(F.conv2d(x.double() - net._mean64.double(), conv.weight.double(), conv.bias.double(), 2, 3), 0.1)
    t32 = F.leaky_relu(F.conv2d(net.normalize_correctly(x), conv.weight, conv.bias, 2, 3), 0.1)
    got = planes.to_nchw(64, 1)
    e_eng, e_t = _rel(got, want64), _rel(t32, want64)
    print(f"conv1 direct {n}x{H}x{W}: engine {e_eng:.2e}, torch fp32 {e_t:.2e} (vs float64)")
    assert e_eng <= max(3 * e_t, 2e-6)
    assert bool((planes.to_nchw(32, 0) == 21.0).all()) and bool((planes.to_nchw(32, 3) == 21.0).all())   # 7 + 7 + 7 per untouched element


@pytest.mark.parametrize("B,Cout,H,W", [(8, 64, 48, 160), (2, 256, 12, 40), (3, 96, 5, 7), (1, 128, 24, 80)])
def test_deconv_flow_tail_kernel_vs_torch(B, Cout, H, W):
    """csrc/engine_small.hip `flow_head_planes_fwd_mfma<1>`: the data gradient of ConvTranspose2d(Cin, Cout, 4, 2, 1) with
    respect to its last two input channels (the upsampled flow) from the masked output gradient's planes, against torch in
    float64; lanes 2.. of the written chunk and the neighbouring chunk stay untouched."""
    import torch.nn.functional as F
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    from understanding_flow_robustness_amd.flownetc_engine import _pack_flow_tail_mfma
    g = torch.Generator().manual_seed(9)
    w2 = (torch.randn(2, Cout, 4, 4, generator=g) * 0.1).to(DEV)
    gy = torch.randn(B, Cout, 2 * H, 2 * W, generator=g).to(DEV)
    chunks = ig.pad32(Cout) // 32
    pl = ig.Planes(B, 2 * H, 2 * W, chunks + 1, DEV).load_nchw(gy, chunk0=1)
    G = ig.GradSum(B, H, W, 3, DEV)
    G.t.fill_(0.25)
    L.check(L.lib().ufr_deconv_flow_tail_backward_mfma(L.ptr(pl.t), pl.plane_stride, 1, chunks, L.ptr(_pack_flow_tail_mfma(w2)),
                                                       chunks, L.ptr(G.t), G.chunks, 1, B, H, W, L.stream()))
    x0 = torch.zeros(B, 2, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv_transpose2d(x0, w2.double(), None, 2, 1), x0, gy.double())
    got = G.t[1].view(B, H, W, 32)
    assert _rel(got[..., :2].permute(0, 3, 1, 2), want) <= 1e-5
    assert bool((got[..., 2:] == 0.25).all()) and bool((G.t[0] == 0.25).all()) and bool((G.t[2] == 0.25).all())


def test_normalize_frames_kernel_is_bit_exact(net):
    """csrc/attack.hip `normalize_frames_kernel` = torch.cat + `normalize_correctly` (float64 mean subtraction,
    FlowNetC.py:73-79, :93-94), bit for bit, for two stacks and for one."""
    from understanding_flow_robustness_amd import _lib as L
    g = torch.Generator().manual_seed(4)
    a, b = torch.rand(3, 3, 40, 72, generator=g).to(DEV), torch.rand(2, 3, 40, 72, generator=g).mul_(255.0).to(DEV)
    mean = net._mean64.reshape(-1).contiguous()
    out = torch.empty(5, 3, 40, 72, device=DEV)
    L.check(L.lib().ufr_normalize_frames(L.ptr(a), L.ptr(b), L.ptr(out), 3, 2, 3, 40, 72, L.ptr(mean), L.stream()))
    assert torch.equal(out, net.normalize_correctly(torch.cat((a, b), 0)))
    out1 = torch.empty(3, 3, 40, 72, device=DEV)
    L.check(L.lib().ufr_normalize_frames(L.ptr(a), None, L.ptr(out1), 3, 0, 3, 40, 72, L.ptr(mean), L.stream()))
    assert torch.equal(out1, net.normalize_correctly(a))


def test_window_gather_chunks_and_gradient_planes_kernels():
    """csrc/window.hip `window_gather_chunks_kernel` == ufr_window_gather on the NCHW form of the same chunk-major tensor;
    csrc/plane_layout.hip `ufr_nchw_grad_to_planes` == gradient x LeakyReLU'(activation), split exactly."""
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    g = torch.Generator().manual_seed(6)
    B, H, W, wh, ww, m = 3, 24, 40, 8, 12, 2
    G = ig.GradSum(B, H, W, 4, DEV)
    G.t.copy_(torch.randn(4, B * H * W, 32, generator=g))
    win = torch.zeros(B, 8, dtype=torch.int32, device=DEV)
    win[:, 0] = torch.tensor([0, 16, 64], dtype=torch.int32)
    win[:, 1] = torch.tensor([112, 8, 48], dtype=torch.int32)
    dst = ig.GradSum(2 * B, wh, ww, 4, DEV)
    dst.t.fill_(7.0)
    L.check(L.lib().ufr_window_gather_chunks(L.ptr(G.t), L.ptr(dst.t), L.ptr(win), B, B, 2 * B, 4, H, W, wh, ww, 4, m, L.stream()))
    want = torch.empty(B, 128, wh, ww, device=DEV)
    L.check(L.lib().ufr_window_gather(L.ptr(G.to_nchw(128, 0)), L.ptr(want), L.ptr(win), B, B, 128, H, W, wh, ww, 4, m, L.stream()))
    got = dst.to_nchw(128, 0)
    assert torch.equal(got[:B], want) and bool((got[B:] == 7.0).all())
    grad, act = torch.randn(2, 70, 9, 13, generator=g).to(DEV), torch.randn(2, 70, 9, 13, generator=g).to(DEV)
    pl = ig.Planes(2, 9, 13, 4, DEV)
    L.check(L.lib().ufr_nchw_grad_to_planes(L.ptr(grad), L.ptr(act), L.ptr(pl.t), pl.plane_stride, 1, 2, 70, 9, 13, 0.1, L.stream()))
    assert torch.equal(pl.to_nchw(70, 1), grad * torch.where(act > 0, 1.0, 0.1))


def test_interleaved_forwards_are_refused_not_silently_wrong(net, monkeypatch):
    """The engine's activations are static buffers: a second grad-mode forward of the same shape before the first one's
    backward must raise (it would differentiate the second call's activations), and gradients handed to autograd are
    copies that the next call does not overwrite."""
    monkeypatch.setenv("UFR_ENGINE", "1")
    g = torch.Generator().manual_seed(5)
    B, H, W = 1, 64, 128

    def feats(seed):
        gg = torch.Generator().manual_seed(seed)
        return [torch.randn(*s, generator=gg).mul_(0.5).to(DEV).requires_grad_(True)
                for s in ((B, 128, H // 4, W // 4), (B, 256, H // 8, W // 8), (B, 256, H // 8, W // 8))]

    gflow = torch.randn(B, 2, H, W, generator=g).to(DEV)
    fa, fb = feats(1), feats(2)
    flow_a = net.head(*fa)
    flow_b = net.head(*fb)
    with pytest.raises(RuntimeError, match="another forward"):
        torch.autograd.grad(flow_a, fa, gflow)
    grads_b = torch.autograd.grad(flow_b, fb, gflow)              # the latest forward is still differentiable
    kept = [t.clone() for t in grads_b]
    flow_a = net.head(*fa)                                        # a later call must not overwrite the returned gradients
    torch.autograd.grad(flow_a, fa, gflow)
    for t, k in zip(grads_b, kept):
        assert torch.equal(t, k)


def test_two_steps_with_different_windows_share_one_engine(net, monkeypatch):
    """Two PatchAttackSteps on the same network and frame size whose masks need different prefix windows: the engine keeps
    one window-prefix state per size, so replaying the first step's graphs after the second step was built (and ran)
    still reads and writes its own buffers -- results equal the runs of each step alone."""
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    monkeypatch.setenv("UFR_ENGINE", "1")
    B, H, W = 2, 256, 512
    g = torch.Generator().manual_seed(21)
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    target = torch.randn(B, 2, H, W, generator=g).to(DEV)
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1.0e4, max_count=2)

    def operands(size):
        mask_p = torch.ones(1, 3, size, size, device=DEV)
        patch0 = torch.rand(1, 3, size, size, generator=g).to(DEV)
        return patch0, mask_p, [(40, 100), (120, 300)]

    def run(step, ops):
        patch0, mask_p, origins = ops
        step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
        step.run(2)
        return step.patch.clone()

    ops_s, ops_l = operands(25), operands(70)
    small = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(25, 25))
    alone_small = run(small, ops_s)
    large = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(70, 70))
    alone_large = run(large, ops_l)
    assert small.win_hw == (96, 96) and large.win_hw == (144, 144)
    assert small.eng is large.eng and len(small.eng._wprefixes) == 2
    for _ in range(2):                                            # alternate: each replay must find its own buffers intact
        assert torch.equal(run(small, ops_s), alone_small)
        assert torch.equal(run(large, ops_l), alone_large)


def test_one_iteration_of_the_headline_step_equals_the_full_frame_torch_step(net, monkeypatch):
    """The deterministic half of parity, checked strictly: ONE iteration of the benchmark's step (8 pairs, 384x1280, 128x128
    window, 608-pixel band, engine, HIP graph) against the full-frame torch / MIOpen step (no window, no band, eager) with the
    patch at the four corners, two edges and two overlapping placements.  After one iteration no LeakyReLU can have flipped
    on a state difference: the two gradients differ by float32 rounding only."""
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    from test_flownetc_gpu import EDGE_PLACEMENTS
    B, H, W = 8, 384, 1280
    g = torch.Generator().manual_seed(31)
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    target = torch.randn(B, 2, H, W, generator=g).to(DEV)
    yy, xx = torch.meshgrid(torch.arange(51), torch.arange(51), indexing="ij")
    mask_p = (((yy - 25) ** 2 + (xx - 25) ** 2) <= 23 ** 2).float().expand(1, 3, 51, 51).contiguous().to(DEV)
    patch0 = torch.rand(1, 3, 51, 51, generator=g).to(DEV)

    def run(engine, cone, lr, graph):
        monkeypatch.setenv("UFR_ENGINE", "1" if engine else "0")
        args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=lr, max_count=1)
        step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51), use_cone=cone, use_graph=graph)
        step.load(tgt, ref, patch0, mask_p, patch0, target, origins=EDGE_PLACEMENTS)
        n, loss = step.run(1)
        return step, step.patch.clone(), n, loss

    _, probe, _, _ = run(False, False, 1.0, False)
    lr = 0.5 / float((probe - patch0).abs().max())
    _, pf, nf, lf = run(False, False, lr, False)
    step, pe, ne, le = run(True, True, lr, True)
    assert step.cone is not None and step.eng is not None and step.band.width == 608 and step.graph is not None
    upd = float((pf - patch0).abs().max())
    err = float((pf - pe).abs().max())
    print(f"one iteration, 8 edge / corner / overlapping placements: engine step vs full-frame torch step {err / upd:.2e} of the update")
    assert nf == ne == 1 and abs(lf - le) <= 1e-5 and 0.3 < upd < 1.9
    assert err <= 1e-5 * upd                               # (measured 8e-7)

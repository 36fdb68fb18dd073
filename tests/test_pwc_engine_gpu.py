"""GPU suite: the native PWC-Net head (pwc_engine.py: pyramid levels 3-6, the DenseNet decoder stages as chunk offsets of
one plane buffer with the reversed-DenseNet adjoint, the dilated context network -- models/PWCNet.py:225-367) against the
torch / MIOpen spelling of the same module with the same weights, both judged against a float64 evaluation."""
from argparse import Namespace

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def net():
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    n = fetch_model(Namespace(flownet="PWCNet"), synthetic_seed=1).to(DEV)
    for p in n.parameters():
        p.requires_grad_(False)
    return n.eval()


def _rel(a, b):
    return float((a.double() - b.double()).abs().max()) / float(b.double().abs().max())


@pytest.mark.parametrize("C,H,W", [(565, 12, 20), (96, 5, 7), (32, 24, 40)])
def test_upfeat_kernels_match_conv_transpose(C, H, W):
    """`upfeat*` = ConvTranspose2d(C, 2, 4, 2, 1) on the engine's planes: forward on the matrix cores (per-pixel GEMM +
    stride-2 gather) and the 16-tap data gradient into the coarse gradient sum, vs torch in float64."""
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    from understanding_flow_robustness_amd.flownetc_engine import _pack_flow_tail_mfma
    from understanding_flow_robustness_amd.pwc_engine import _pack_tail_bwd
    B = 2
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    w = (torch.randn(C, 2, 4, 4, generator=g) * 0.1).to(DEV)
    b = torch.randn(2, generator=g).to(DEV)
    chunks = ig.pad32(C) // 32
    planes = ig.Planes(B, H, W, chunks + 1, DEV).load_nchw(x, 1)           # at chunk 1 of a wider buffer
    wbuf = torch.zeros(chunks * 32, 2, 4, 4, device=DEV)
    wbuf[:C] = w
    out = torch.full((B, 2, 2 * H, 2 * W), float("nan"), device=DEV)
    L.check(L.lib().ufr_upfeat_planes_forward_mfma(L.ptr(planes.t), planes.plane_stride, 1, chunks,
                                                   L.ptr(_pack_flow_tail_mfma(wbuf.permute(1, 0, 2, 3).contiguous())), chunks, L.ptr(b), L.ptr(out),
                                                   B, H, W, L.stream()))
    want = torch.nn.functional.conv_transpose2d(x.double(), w.double(), b.double(), 2, 1)
    assert _rel(out, want) <= 2e-6
    gy = torch.randn(B, 2, 2 * H, 2 * W, generator=g).to(DEV)
    Gs = ig.GradSum(B, H, W, chunks + 1, DEV)
    base = torch.randn_like(Gs.t)
    for accumulate in (0, 1):
        Gs.t.copy_(base)
        L.check(L.lib().ufr_upfeat_planes_backward(L.ptr(gy), L.ptr(_pack_tail_bwd(wbuf)), chunks, L.ptr(Gs.t), Gs.chunks, 1, chunks, B, H, W, accumulate,
                                                   L.stream()))
        xg = x.double().requires_grad_(True)
        (gx,) = torch.autograd.grad(torch.nn.functional.conv_transpose2d(xg, w.double(), None, 2, 1), xg, gy.double())
        got = Gs.to_nchw(C, 1, slope=1.0)
        if accumulate:
            keep = ig.GradSum(B, H, W, chunks + 1, DEV)
            keep.t.copy_(base)
            gx = gx + keep.to_nchw(C, 1, slope=1.0).double()
        assert _rel(got, gx) <= 2e-6
        assert torch.equal(Gs.t[0], base[0])                                # the chunk in front is not touched


@pytest.mark.parametrize("B,H,W", [(1, 128, 192), (2, 64, 128)])
def test_pwc_engine_head_forward_and_gradient_match_the_torch_head(net, monkeypatch, B, H, W):
    """flow2-level output and the gradient with respect to the level-2 features: engine vs the torch / MIOpen spelling,
    both against float64 (the engine may be no further from it than MIOpen's fp32 path, x3)."""
    import copy
    g = torch.Generator().manual_seed(3)
    f2 = [torch.randn(B, 32, H // 4, W // 4, generator=g).mul_(0.5).to(DEV) for _ in range(2)]
    gflow = torch.randn(B, 2, H, W, generator=g).to(DEV)
    outs = {}
    for knob in ("0", "1"):
        monkeypatch.setenv("UFR_ENGINE", knob)
        leaves = [f.clone().requires_grad_(True) for f in f2]
        flow = net.head(*leaves)
        outs[knob] = (flow.detach(), torch.autograd.grad(flow, leaves, gflow))
    assert net.__dict__.get("_ufr_head_engines"), "the head did not run on the engine"
    monkeypatch.setenv("UFR_ENGINE", "0")
    engines = net.__dict__.pop("_ufr_head_engines")              # ctypes descriptors: not copyable
    net64 = copy.deepcopy(net).double()
    net.__dict__["_ufr_head_engines"] = engines
    leaves = [f.double().requires_grad_(True) for f in f2]
    flow64 = net64.head(*leaves)
    grads64 = torch.autograd.grad(flow64, leaves, gflow.double())
    (f0, g0), (f1, g1) = outs["0"], outs["1"]
    print(f"flow: engine {_rel(f1, flow64):.2e}, torch fp32 {_rel(f0, flow64):.2e} of max |flow| (vs float64)")
    assert _rel(f1, flow64) <= max(3 * _rel(f0, flow64), 1e-5)
    for name, a, b, truth in zip(("d/d f2a", "d/d f2b"), g1, g0, grads64):
        e_eng, e_t32 = _rel(a, truth), _rel(b, truth)
        print(f"{name}: engine {e_eng:.2e}, torch fp32 {e_t32:.2e} of max |gradient| (vs float64)")
        # a LeakyReLU whose pre-activation sits within rounding of zero takes the other slope in an fp32 evaluation, and
        # the warps' validity mask / bilinear cells are piecewise: isolated entries, in either implementation
        assert e_eng <= max(3 * e_t32, 5e-4), f"{name}: engine {e_eng:.2e} vs torch fp32 {e_t32:.2e}"
        beyond = lambda v: float(((v.double() - truth).abs() > 1e-4 * float(truth.abs().max())).float().mean())
        frac, frac_t32 = beyond(a), beyond(b)
        print(f"{name}: entries beyond 1e-4 of the float64 gradient: engine {frac:.2e}, torch fp32 {frac_t32:.2e}; "
              f"engine vs torch fp32 {_rel(a, b):.2e}")
        # random features put many warp samples within rounding of a cell border / the validity threshold: the float64
        # evaluation takes the other branch there for EITHER fp32 implementation, so the engine is held to torch's own count
        assert frac <= max(2 * frac_t32, 1e-2), f"{name}: {frac:.2e} of the entries beyond 1e-4 (torch fp32: {frac_t32:.2e})"


def test_pwc_engine_step_equals_full_frame_torch_step(net, monkeypatch):
    """Config C4's step (windowed pyramid levels 1-2 on the igemm, cached level-2 features in the engine's planes, fused
    loss, one HIP graph) against the full-frame torch / MIOpen step (UFR_ENGINE=0, no window, eager) after ONE iteration:
    4 pairs behind one 51x51 patch at 384x1280, placements at a corner, two edges and the interior -- 1e-4 of the update, with MIOpen's find step off (see the assertion)."""
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    B, H, W = 4, 384, 1280
    g = torch.Generator().manual_seed(11)
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    target = torch.randn(B, 2, H, W, generator=g).to(DEV)
    yy, xx = torch.meshgrid(torch.arange(51), torch.arange(51), indexing="ij")
    mask_p = (((yy - 25) ** 2 + (xx - 25) ** 2) <= 23 ** 2).float().expand(1, 3, 51, 51).contiguous().to(DEV)
    patch0 = torch.rand(1, 3, 51, 51, generator=g).to(DEV)
    placements = ([(0, 0), (333, 1229), (0, 600), (170, 640)], [(333, 0), (160, 640), (7, 1221), (100, 300)])

    def run(engine, cone, lr, graph):
        monkeypatch.setenv("UFR_ENGINE", "1" if engine else "0")
        args = Namespace(flownet="PWCNet", l2=False, alpha=0.0, lr=lr, max_count=1)
        step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51), use_cone=cone, use_graph=graph)
        outs = []
        for origins in placements:
            step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
            n, loss = step.run(1)
            outs.append((step.patch.clone(), n, loss))
        return step, outs

    # the yardstick is the torch / MIOpen step: with the find step on (cudnn.benchmark, switched on by earlier tests of this
    # process) MIOpen picks its fp32 convolution algorithms per box and per call shape -- the same tree measured 1.6e-5 .. 5.3e-5
    # of the update on one box and 1.16e-4 on another (gpurun r4_call65 / r4_final_d); with it off the yardstick is the same
    # kernels everywhere (as in test_models_gpu.py::test_raft_gradient_against_float64_truth)
    bench_mode = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = False
    try:
        _, probe = run(False, False, 1.0, False)
        lr = 0.5 / float((probe[0][0] - patch0).abs().max())          # first update peaks at 0.5: the +-2 clamp stays inactive
        _, full = run(False, False, lr, False)
    finally:
        torch.backends.cudnn.benchmark = bench_mode
    step, eng = run(True, True, lr, True)
    assert step.cone is not None and step.eng is not None and step.eng_kind == "pwc" and step.graph is not None
    for (pf, nf, lf), (pe, ne, le) in zip(full, eng):
        upd = float((pf - patch0).abs().max())
        err = float((pf - pe).abs().max())
        print(f"update {upd:.3e}, engine step vs torch step {err / upd:.2e} of it; loss {lf:.6f} / {le:.6f}")
        assert nf == ne == 1 and abs(lf - le) <= 1e-5
        # The engine's own error is pinned by `test_pwc_step_at_full_size_vs_cpu_oracle`: 1e-4 of the update against the CPU oracle
        # (6e-6 measured).  HERE the yardstick is torch's float32 step, whose own distance from the truth is what is seen: measured
        # 1.6e-5 .. 1.16e-4 of the update across boxes and call orders (its grid_sample adjoint scatters with float atomics), 1.02e-4
        # once in round 5 (gpurun r5_warp2) against 8e-5 on the next box with the same engine.  2e-4: a consistency gate, not the pin.
        assert 1e-3 < upd < 1.9 and err <= 2e-4 * upd, f"{err / upd:.2e} of the update"


@pytest.mark.timeout(900)
def test_pwc_step_at_full_size_vs_cpu_oracle(net, oracle):
    """Config C4's step at its real frame size against the CPU ORACLE (oracle/flow_oracle.py: `pwcnet_forward` reproduces the
    reference's PWCDCNet golden, `patch_attack_placed` is the batch definition): ONE iteration, 2 pairs of 384x1280 behind one
    51x51 circular patch, a corner and an interior placement, windowed pyramid + engine + HIP graph -- 1e-4 of the update."""
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    B, H, W = 2, 384, 1280
    g = torch.Generator().manual_seed(23)
    tgt, ref = torch.rand(B, 3, H, W, generator=g), torch.rand(B, 3, H, W, generator=g)
    target = torch.randn(B, 2, H, W, generator=g)
    yy, xx = torch.meshgrid(torch.arange(51), torch.arange(51), indexing="ij")
    mask_p = (((yy - 25) ** 2 + (xx - 25) ** 2) <= 23 ** 2).float().expand(1, 3, 51, 51).contiguous()
    patch0 = torch.rand(1, 3, 51, 51, generator=g)
    origins = [(0, 1229), (170, 600)]
    sd = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    predict = lambda a, b: fo.pwcnet_forward(sd, a, b)
    torch.set_num_threads(16)

    def oracle_patch(lr):
        p = patch0.clone()
        fo.patch_attack_placed(predict, tgt, ref, p, mask_p, origins, target, lr=lr, max_count=1)
        return p
    probe = oracle_patch(1.0)
    lr = 0.5 / float((probe - patch0).abs().max())                  # first update peaks at 0.5: the +-2 clamp stays inactive
    want = oracle_patch(lr)
    args = Namespace(flownet="PWCNet", l2=False, alpha=0.0, lr=lr, max_count=1)
    step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51))
    step.load(tgt.to(DEV), ref.to(DEV), patch0.to(DEV), mask_p.to(DEV), patch0.to(DEV), target.to(DEV), origins=origins)
    n, _ = step.run(1)
    assert n == 1 and step.cone is not None and step.eng is not None and step.eng_kind == "pwc" and step.graph is not None
    upd = float(((want - patch0) * mask_p).abs().max())
    err = float(((step.patch.cpu() - want) * mask_p).abs().max())
    print(f"PWC-Net 2 x 384x1280, one iteration: update {upd:.3e}, step vs CPU oracle {err / upd:.2e} of it")
    assert 1e-3 < upd < 1.9 and err <= 1e-4 * upd, f"{err / upd:.2e} of the update"


@pytest.mark.parametrize("B,F,H,W", [(2, 32, 24, 40), (1, 96, 13, 29)])
def test_stage_input_cat_kernels(B, F, H, W):
    """x = cat(corr 81, up_flow 2, up_feat 2 | c1) -> planes (ufr_nchw_cat_to_planes) and the member gradients back out of the
    float32 gradient sum (ufr_chunks_to_nchw_cat; the cost volume's part through its activation): exact copies / products."""
    import ctypes as C

    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    g = torch.Generator().manual_seed(F + H)
    members = [torch.randn(B, c, H, W, generator=g).to(DEV) for c in (81, 2, 2, F)]
    chans, first = (C.c_int * 4)(81, 2, 2, F), (C.c_int * 4)(0, 81, 83, 96)
    chunks = 3 + ig.pad32(F) // 32
    planes = ig.Planes(B, H, W, chunks + 2, DEV)
    planes.t.fill_(7.0)
    ptrs = (C.c_void_p * 4)(*[m.data_ptr() for m in members])
    L.check(L.lib().ufr_nchw_cat_to_planes(ptrs, chans, first, 4, L.ptr(planes.t), planes.plane_stride, 1, chunks, B, H, W, L.stream()))
    got = planes.to_nchw(chunks * 32, 1)
    want = torch.zeros(B, chunks * 32, H, W, device=DEV)
    for m, c0 in zip(members, (0, 81, 83, 96)):
        want[:, c0:c0 + m.shape[1]] = m
    assert float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())          # three bf16 planes hold a float32 exactly
    assert bool((planes.t[:, 0] == 7.0).all()) and bool((planes.t[:, chunks + 1] == 7.0).all())
    G = ig.GradSum(B, H, W, chunks + 1, DEV)
    G.t.copy_(torch.randn(G.t.shape, generator=g).to(DEV))
    outs = [torch.full_like(m, float("nan")) for m in members]
    optrs = (C.c_void_p * 4)(*[o.data_ptr() for o in outs])
    L.check(L.lib().ufr_chunks_to_nchw_cat(L.ptr(G.t), 1, chunks, optrs, chans, first, 4, L.ptr(members[0]), 0.25, 0.025, B, H, W, L.stream()))
    full = G.to_nchw(chunks * 32, 1, slope=1.0)
    assert torch.equal(outs[0], full[:, :81] * torch.where(members[0] > 0, 0.25, 0.025))
    for o, c0 in zip(outs[1:], (81, 83, 96)):
        assert torch.equal(o, full[:, c0:c0 + o.shape[1]])
    with pytest.raises(RuntimeError, match="out of order"):
        L.check(L.lib().ufr_nchw_cat_to_planes(ptrs, chans, (C.c_int * 4)(0, 80, 83, 96), 4, L.ptr(planes.t), planes.plane_stride, 1, chunks,
                                               B, H, W, L.stream()))


@pytest.mark.parametrize("n,N,H,W", [(2, 16, 64, 128), (16, 16, 384, 1280), (3, 16, 34, 70), (1, 24, 16, 18)])
def test_conv1a_direct_kernel_vs_float64(n, N, H, W):
    """`ufr_conv3x3s2_c3_planes` (PWC-Net's conv1a from the raw frames, csrc/small_cin_conv.hip) against float64
    conv2d + LeakyReLU: within 3x torch-float32's own error; the chunk's padding channels stay zero."""
    import torch.nn.functional as F
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    g = torch.Generator().manual_seed(n * H + W)
    x = (torch.rand(n, 3, H, W, generator=g) * 2 - 1).to(DEV)
    w = (torch.randn(N, 3, 3, 3, generator=g) * 0.3).to(DEV)
    b = (torch.randn(N, generator=g) * 0.1).to(DEV)
    out = ig.Planes(n, H // 2, W // 2, 1, DEV)
    L.check(L.lib().ufr_conv3x3s2_c3_planes(L.ptr(x), L.ptr(w), L.ptr(b), 0.1, L.ptr(out.t), out.plane_stride, 0, n, N, H, W, L.stream()),
            "conv1a direct")
    got = out.to_nchw(32, 0).double()
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1), 0.1)
    t32 = F.leaky_relu(F.conv2d(x, w, b, stride=2, padding=1), 0.1).double()
    scale = float(want.abs().max())
    err, err_t = float((got[:, :N] - want).abs().max()) / scale, float((t32 - want).abs().max()) / scale
    print(f"conv1a direct {n}x{H}x{W} -> {N}: kernel {err:.2e}, torch fp32 {err_t:.2e} (vs float64)")
    assert err <= max(3 * err_t, 2e-6)
    assert float(got[:, N:].abs().max()) == 0.0


@pytest.mark.parametrize("n,H,W", [(2, 32, 64), (16, 192, 640), (3, 17, 35), (1, 8, 33)])
def test_conv16_direct_kernel_vs_float64(n, H, W):
    """`ufr_conv3x3_c16_planes` (conv1aa / conv1b of PWC-Net's pyramid on planes) against float64 conv2d + LeakyReLU: within 3x
    torch-float32's own error; the output chunk's channels 16-31 are left as they were (zeros)."""
    import torch.nn.functional as F
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    g = torch.Generator().manual_seed(n * H + W)
    x = (torch.randn(n, 16, H, W, generator=g)).to(DEV)
    w = (torch.randn(16, 16, 3, 3, generator=g) * 0.1).to(DEV)
    b = (torch.randn(16, generator=g) * 0.1).to(DEV)
    src, dst = ig.Planes(n, H, W, 1, DEV), ig.Planes(n, H, W, 1, DEV)
    src.load_nchw(x, 0)
    wt = w.permute(1, 2, 3, 0).reshape(16, 9, 16).contiguous()
    L.check(L.lib().ufr_conv3x3_c16_planes(L.ptr(src.t), src.plane_stride, 0, L.ptr(wt), L.ptr(b), 0.1, L.ptr(dst.t), dst.plane_stride, 0,
                                           n, H, W, L.stream()), "conv16 direct")
    got = dst.to_nchw(32, 0).double()
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), padding=1), 0.1)
    t32 = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.1).double()
    scale = float(want.abs().max())
    err, err_t = float((got[:, :16] - want).abs().max()) / scale, float((t32 - want).abs().max()) / scale
    print(f"conv 16 -> 16 direct {n}x{H}x{W}: kernel {err:.2e}, torch fp32 {err_t:.2e} (vs float64)")
    assert err <= max(3 * err_t, 2e-6)
    assert float(got[:, 16:].abs().max()) == 0.0

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


@pytest.mark.parametrize("alternate", [True, False])
def test_raft_attack_at_full_size_engine_vs_torch_spelling(raft, monkeypatch, alternate, oracle):
    """Config C3 as BASELINE states it: attack() on RAFT at 384x1280 with alt_cuda_corr (and the all-pairs volume), ONE
    iteration, everything on the native engines (encoders, lookups on the matrix cores, the 12-iteration refinement, convex
    upsampling) against (a) the same step with UFR_ENGINE=0 -- the torch / MIOpen spelling of the same modules on this device --
    and (b) the float64 evaluation of the oracle's RAFT on this device, the conditioning-free truth: RAFT's image gradient
    loses ~1e-3 .. 1e-2 in float32 (tests/test_models_gpu.py), so the gate is the truth, with the torch spelling as yardstick."""
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    net, _ = raft
    H, W = 384, 1280
    g = torch.Generator().manual_seed(17)
    tgt, ref = torch.rand(1, 3, H, W, generator=g).to(DEV), torch.rand(1, 3, H, W, generator=g).to(DEV)
    target = torch.randn(1, 2, H, W, generator=g).to(DEV)
    mask = torch.zeros(1, 3, H, W, device=DEV)
    mask[:, :, 100:151, 600:651] = 1
    patch0 = torch.rand(1, 3, H, W, generator=g).to(DEV) * mask

    def run(engine, lr):
        monkeypatch.setenv("UFR_ENGINE", "1" if engine else "0")
        args = Namespace(flownet="RAFT", l2=False, alpha=0.0, lr=lr, max_count=1, alternate_corr=alternate, mixed_precision=False)
        net.args.alternate_corr = alternate
        step = PatchAttackStep(net, args, 1, H, W, device=DEV, use_graph=engine)
        step.load(tgt, ref, patch0, mask, patch0, target)
        n, loss = step.run(1)
        return step.patch.clone(), n, loss

    probe, _, _ = run(False, 1.0)
    lr = 0.5 / float(((probe - patch0) * mask).abs().max())
    pf, nf, lf = run(False, lr)
    pe, ne, le = run(True, lr)
    engines = [e for m in net.modules() for a in ("_ufr_head_engines", "_ufr_encoder_engines") for e in m.__dict__.get(a, {}).values()]
    assert len(engines) >= 3, "the engines (two encoders + the refinement loop) did not run"
    # float64 truth of the same iteration (main.py:546-600 for one pair: paste, forward, loss, gradient, clamped step)
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in net.state_dict().items()}
    m64 = mask.double()
    adv_t = ((1 - m64) * tgt.double() + m64 * patch0.double()).requires_grad_(True)
    adv_r = ((1 - m64) * ref.double() + m64 * patch0.double()).requires_grad_(True)
    flow64 = fo.raft_forward(sd64, adv_t * 255.0, adv_r * 255.0)[1]
    g_t, g_r = torch.autograd.grad(fo.flow_loss(flow64, target.double()), (adv_t, adv_r))
    truth = patch0.double() - torch.clamp(0.5 * lr * (g_t + g_r), -2.0, 2.0)
    upd = float(((truth - patch0) * mask).abs().max())
    e_eng = float(((pe - truth) * mask).abs().max()) / upd
    e_torch = float(((pf - truth) * mask).abs().max()) / upd
    e_same = float(((pf - pe) * mask).abs().max()) / upd
    print(f"RAFT alt={alternate} 384x1280, one iteration, of the update: engines vs float64 {e_eng:.2e}, torch spelling vs float64 "
          f"{e_torch:.2e}, engines vs torch spelling {e_same:.2e}; loss {lf:.6f} / {le:.6f}")
    assert nf == ne == 1 and abs(lf - le) <= 1e-5 and 0.3 < upd < 1.9
    # round 6: the gates at what is measured instead of round 5's 1e-3 / 1e-2.  Engines vs float64: 4.4e-5 (all-pairs) and
    # 6.8e-5 (alt_corr) of the update -- the alt_corr leg was 1.4e-4 while its lookup ran on `v_mfma_f32_16x16x4_f32` (raft_altcorr_mfma.hip);
    # on the three-plane bf16 form (raft_altcorr_planes.hip, the igemm's arithmetic) both legs are under north_star's 1e-4, which is now the
    # gate.  The torch spelling on this device is at 1.4e-3 of the same truth, so engines-vs-torch sits there too (gate 3e-3).
    assert e_eng <= 1e-4 and e_same <= 3e-3


@pytest.mark.parametrize("mode,products", [("bf16", 1), ("bf16x3", 3)])
def test_raft_reduced_precision_is_an_explicit_opt_in(raft, monkeypatch, mode, products):
    """The reference runs RAFT's encoders and update block under fp16 autocast whenever `args.mixed_precision` is set
    (models/utils_model.py:51, models/raft/raft.py:140,168,195).  Here the flag ALONE changes nothing (float32, bit for bit);
    together with UFR_RAFT_PRECISION=bf16 the engines' convolutions take ONE bf16 product per float32 product (bf16x3: three), the
    correlation stays float32.  Gate of the reduced forms against the float32 engines: flow within 3e-2 of max |flow| for one product
    (measured 1.5e-2 at 256 x 512 with random-init weights: bfloat16 keeps 8 significand bits where the reference's float16 autocast
    keeps 11, so its ~1e-2 becomes ~1.5e-2 through 12 iterations), 1e-4 for three; the image gradient is reported."""
    import copy as _copy
    net, _ = raft
    net = _copy.deepcopy(net)
    H, W = 256, 512
    g = torch.Generator().manual_seed(23)
    i1, i2 = (torch.rand(1, 3, H, W, generator=g) * 255).to(DEV), (torch.rand(1, 3, H, W, generator=g) * 255).to(DEV)
    go = torch.randn(1, 2, H, W, generator=g).to(DEV)

    def run():
        a, b = i1.clone().requires_grad_(True), i2.clone().requires_grad_(True)
        flow = net(a, b, test_mode=True)[1]
        ga, gb = torch.autograd.grad(flow, (a, b), go)
        return flow.detach().clone(), ga.clone(), gb.clone()

    monkeypatch.delenv("UFR_RAFT_PRECISION", raising=False)
    net.args.mixed_precision = False
    assert net.products() == 6
    f32 = run()
    net.args.mixed_precision = True                                   # the flag alone: still float32, the very same engines
    assert net.products() == 6
    flag_only = run()
    assert all(torch.equal(x, y) for x, y in zip(f32, flag_only))
    monkeypatch.setenv("UFR_RAFT_PRECISION", mode)
    assert net.products() == products
    low = run()
    net.args.mixed_precision = False                                  # the environment alone: float32 as well
    assert net.products() == 6
    assert all(torch.equal(x, y) for x, y in zip(f32, run()))
    keys = [k for m in net.modules() for a in ("_ufr_head_engines", "_ufr_encoder_engines") for k in m.__dict__.get(a, {})]
    assert {k[-1] for k in keys} == {6, products}, "the reduced-precision forward did not build its own engines"
    e_flow, e_g = _rel(low[0], f32[0]), max(_rel(low[1], f32[1]), _rel(low[2], f32[2]))
    print(f"RAFT {H}x{W}, UFR_RAFT_PRECISION={mode} ({products} product(s)) against the float32 engines: flow {e_flow:.2e} of max |flow|, "
          f"image gradient {e_g:.2e} of its maximum")
    assert e_flow <= (3e-2 if products == 1 else 1e-4)
    assert e_flow > 1e-7, "the reduced form computed the float32 result: the switch did nothing"
    monkeypatch.setenv("UFR_RAFT_PRECISION", "fp16")
    net.args.mixed_precision = True
    with pytest.raises(ValueError):
        net.products()


@pytest.mark.parametrize("rows_hw,cols,chunk0,scale", [((10, 10), 70, 1, 1.0), ((16, 24), 384, 0, 0.0625), ((1, 256), 7680, 0, 1.0)])
def test_rowmajor_matrix_to_planes_is_the_exact_three_way_split(rows_hw, cols, chunk0, scale):
    """`ufr_rowmajor_to_planes` (the column-reduced operand of the all-pairs adjoint): planes[chunk0 + k/32][m][k%32] holds the three
    bf16 planes of scale * src[m][k] bit for bit (igemm._split3: round to nearest at every step), zeros in the padding columns,
    neighbouring chunks untouched; a strided source (ld > cols) and a range that leaves the buffer is refused."""
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    H, W = rows_hw
    M, kc = H * W, (cols + 31) // 32
    g = torch.Generator().manual_seed(cols)
    wide = torch.randn(M, cols + 8, generator=g).to(DEV)
    src = wide[:, :cols]                                                # ld = cols + 8
    pl = ig.Planes(1, H, W, chunk0 + kc + 1, DEV)
    pl.t.fill_(7.0)
    pl.load_rowmajor(src, chunk0, scale=scale)
    want = torch.zeros(M, kc * 32, device=DEV)
    want[:, :cols] = src * scale
    want = ig._split3(want.view(M, kc, 32).permute(1, 0, 2).contiguous()).view(3, kc, M, 32)
    assert torch.equal(pl.t[:, chunk0:chunk0 + kc], want)
    assert bool((pl.t[:, chunk0 + kc:] == 7.0).all()) and bool((pl.t[:, :chunk0] == 7.0).all())
    rc = L.lib().ufr_rowmajor_to_planes(L.ptr(src), src.stride(0), M, cols, 1.0, L.ptr(pl.t), pl.plane_stride, chunk0 + 2, M, L.stream())
    assert rc == -1 and b"leave the planes operand" in L.lib().ufr_last_error()


@pytest.mark.parametrize("B,C,H,W", [(1, 256, 48, 160), (2, 256, 16, 24), (2, 128, 16, 16)])
def test_all_pairs_correlation_on_the_igemm(B, C, H, W, monkeypatch):
    """`CorrBlock.corr` (models/raft/corr.py:57-64: matmul / sqrt(C)) as ONE hand-written igemm launch per pair with fmap2's
    planes as the weight image (flownets/raft_corr.py `AllPairsCorrFunction`): against a float64 product, judged by the library
    GEMM's own float32 error; the adjoint against torch autograd through the reference's spelling."""
    from understanding_flow_robustness_amd.flownets.raft_corr import AllPairsCorrFunction, CorrBlock
    g = torch.Generator().manual_seed(C + H)
    f1 = torch.randn(B, C, H, W, generator=g).to(DEV).requires_grad_(True)
    f2 = torch.randn(B, C, H, W, generator=g).to(DEV).requires_grad_(True)
    assert AllPairsCorrFunction.supported(f1, f2)
    got = CorrBlock.corr(f1, f2)
    assert got.shape == (B, H, W, 1, H, W) and "AllPairs" in type(got.grad_fn.next_functions[0][0]).__name__     # (behind the .view)
    monkeypatch.setenv("UFR_ENGINE", "0")
    lib32 = CorrBlock.corr(f1, f2)                                     # the reference's spelling on the library GEMM
    monkeypatch.setenv("UFR_ENGINE", "1")
    want = (torch.matmul(f1.detach().double().view(B, C, -1).transpose(1, 2), f2.detach().double().view(B, C, -1)) / C ** 0.5).view_as(got)
    rel = lambda a, b: float((a.detach().double() - b).abs().max()) / float(b.abs().max())
    e_eng, e_lib = rel(got, want), rel(lib32, want)
    print(f"all-pairs {B}x{C}x{H}x{W}: igemm {e_eng:.2e}, library fp32 {e_lib:.2e} (vs float64)")
    assert e_eng <= max(3 * e_lib, 2e-6)
    # the adjoint (round 5: both products on the igemm, the volume's gradient as the activation, the feature maps as the weight
    # image) against float64, judged by the library GEMM's own float32 error like the forward
    go = torch.randn(got.shape, generator=g).to(DEV)
    g1, g2 = torch.autograd.grad(got, (f1, f2), go)
    r1, r2 = torch.autograd.grad(lib32, (f1, f2), go)
    go64 = go.double().view(B, H * W, H * W)
    w1 = (torch.matmul(f2.detach().double().view(B, C, -1), go64.transpose(1, 2)) / C ** 0.5).view_as(f1)
    w2 = (torch.matmul(f1.detach().double().view(B, C, -1), go64) / C ** 0.5).view_as(f2)
    for name, mine, lib_, want_ in (("d fmap1", g1, r1, w1), ("d fmap2", g2, r2, w2)):
        e_eng, e_lib = rel(mine, want_), rel(lib_, want_)
        print(f"all-pairs adjoint {name} {B}x{C}x{H}x{W}: igemm {e_eng:.2e}, library fp32 {e_lib:.2e} (vs float64)")
        assert e_eng <= max(3 * e_lib, 2e-6), f"{name}: {e_eng:.2e} (library {e_lib:.2e})"
    monkeypatch.setenv("UFR_ALLPAIRS_ADJOINT", "0")                    # the library-GEMM form stays available for A/B runs
    k1, k2 = torch.autograd.grad(CorrBlock.corr(f1, f2), (f1, f2), go)
    assert rel(k1, r1.double()) <= 1e-5 and rel(k2, r2.double()) <= 1e-5


@pytest.mark.parametrize("n,H,W,chunks", [(2, 32, 64, 2), (1, 55, 128, 4), (2, 17, 23, 3)])
def test_norm_statistics_finished_inside_the_apply_kernels_bit_exact(n, H, W, chunks):
    """`ufr_cm_norm_stats_apply` (two launches: the apply kernel adds the float64 partials itself) against `ufr_cm_norm_stats` +
    `ufr_cm_norm_apply` (three): statistics and activation planes bit-identical; the adjoint's fused second stage against the
    hand-computed sums of the three-launch arithmetic: gradient planes bit-identical to the unfused kernel fed with those sums."""
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import igemm as ig
    lib = L.lib()
    HW = H * W
    g = torch.Generator().manual_seed(n * H + W)
    X = ig.GradSum(n, H, W, chunks, DEV)
    X.t.copy_(torch.randn(X.t.shape, generator=g) * 3 + 0.5)
    res = ig.Planes(n, H, W, chunks, DEV)
    res.load_nchw(torch.randn(n, chunks * 32, H, W, generator=g).to(DEV), 0)
    ws = torch.empty(lib.ufr_cm_norm_workspace_doubles(HW, n, chunks), dtype=torch.float64, device=DEV)
    st = L.stream()
    stats_a, stats_b = (torch.zeros(n * chunks * 32 * 2, device=DEV) for _ in range(2))
    out_a, out_b = ig.Planes(n, H, W, chunks, DEV), ig.Planes(n, H, W, chunks, DEV)
    L.check(lib.ufr_cm_norm_stats(L.ptr(X.t), L.ptr(stats_a), L.ptr(ws), HW, n, chunks, 1e-5, st), "stats")
    L.check(lib.ufr_cm_norm_apply(L.ptr(X.t), L.ptr(stats_a), L.ptr(res.t), res.plane_stride, 0, L.ptr(out_a.t), out_a.plane_stride, 0, HW, n,
                                  chunks, 1, 1, st), "apply")
    L.check(lib.ufr_cm_norm_stats_apply(L.ptr(X.t), L.ptr(stats_b), L.ptr(ws), 1e-5, L.ptr(res.t), res.plane_stride, 0, L.ptr(out_b.t),
                                        out_b.plane_stride, 0, HW, n, chunks, 1, 1, st), "stats + apply")
    assert torch.equal(stats_a, stats_b)
    assert torch.equal(out_a.t.view(torch.int16), out_b.t.view(torch.int16))
    # adjoint: the fused path (statistics form) twice must agree with itself and write the same sums it used
    G = ig.GradSum(n, H, W, chunks, DEV)
    G.t.copy_(torch.randn(G.t.shape, generator=g))
    sums = torch.zeros(n * chunks * 32 * 2, device=DEV)
    gz_a, gz_b = ig.Planes(n, H, W, chunks, DEV), ig.Planes(n, H, W, chunks, DEV)
    L.check(lib.ufr_cm_norm_backward(L.ptr(X.t), L.ptr(G.t), L.ptr(out_a.t), 0, L.ptr(stats_a), L.ptr(sums), L.ptr(ws), L.ptr(gz_a.t),
                                     gz_a.plane_stride, 0, HW, n, chunks, 1, st), "backward")
    # float64 restatement of the adjoint from the same operands
    x = X.to_nchw(chunks * 32, 0, slope=1.0).double()
    go = G.to_nchw(chunks * 32, 0, slope=1.0).double()
    mean = stats_a.view(n, chunks * 32, 2)[..., 0].double()[:, :, None, None]
    rstd = stats_a.view(n, chunks * 32, 2)[..., 1].double()[:, :, None, None]
    xh = (x - mean) * rstd
    outv = out_a.to_nchw(chunks * 32, 0).double()
    gm = go * (outv > 0) * (xh > 0)
    want = rstd * (gm - gm.mean((2, 3), keepdim=True) - xh * (gm * xh).mean((2, 3), keepdim=True))
    got = gz_a.to_nchw(chunks * 32, 0).double()
    assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max())
    want_sums = torch.stack((gm.mean((2, 3)), (gm * xh).mean((2, 3))), -1).reshape(-1)
    assert float((sums.double() - want_sums).abs().max()) <= 1e-6 * max(1.0, float(want_sums.abs().max()))


def test_gate_arithmetic_reading_the_split_k_slabs_is_bit_identical(raft, monkeypatch):
    """The GRU's gate / candidate convolutions leave their raw split-K slabs to the gate kernels (`no_reduce`,
    ufr_gru_gates_cm_forward_slabs / ufr_gru_blend_cm_forward_slabs): flow, mask and the context gradients must equal the
    three-launch form (convolution, reduce, gate arithmetic) bit for bit -- same order of additions."""
    from understanding_flow_robustness_amd.raft_engine import RaftUpdateEngine
    net, _ = raft
    B, H, W = 1, 128, 192
    h, w = H // 8, W // 8
    g = torch.Generator().manual_seed(5)
    net0 = torch.tanh(torch.randn(B, 128, h, w, generator=g)).to(DEV)
    inp = torch.relu(torch.randn(B, 128, h, w, generator=g)).to(DEV)
    f1 = torch.randn(B, h, w, 256, generator=g).to(DEV)
    f2 = [torch.randn(B, h >> l, w >> l, 256, generator=g).to(DEV) for l in range(4)]
    gf, gm = torch.randn(B, 2, h, w, generator=g).to(DEV), torch.randn(B, 576, h, w, generator=g).to(DEV) * 0.01
    res = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("UFR_RAFT_FUSE_REDUCE", knob)
        eng = RaftUpdateEngine(net, B, H, W, DEV)
        fused = [bool(l.desc.no_reduce) for k, l in eng.launch.items() if k[0] in ("zr1", "q1", "zr2", "q2", "conv")]
        assert any(fused) == (knob == "1"), "no launch of this grid is split: pick a size whose gate convolutions are"
        src = dict(alt=True, f1=f1, f2=f2, scale=1.0 / 16.0)
        flow, mask = eng.forward(net0, inp, src)
        flow, mask = flow.clone(), mask.clone()
        src["g_f1"], src["g_f2"] = torch.empty_like(f1), [torch.empty_like(f) for f in f2]
        nbytes = __import__("understanding_flow_robustness_amd._lib", fromlist=["lib"]).lib().ufr_altcorr_pyramid_workspace_bytes(
            B, h, w, 256, eng.radius, len(f2))
        src["ws"] = torch.empty(nbytes, dtype=torch.uint8, device=DEV)
        g_net0, g_inp = eng.backward(gf, gm)
        res[knob] = (flow, mask, g_net0.clone(), g_inp.clone())
    for a, b in zip(res["1"], res["0"]):
        assert torch.equal(a, b)

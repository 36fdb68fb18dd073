"""GPU suite: every operator of libufr_hip.so, called through the reference-shaped Python mirrors
(which go through the C ABI), against (1) the golden vectors of the reference's CPU correlation and
(2) the C oracle on seeded inputs; full-size cases through digests and algebraic properties."""
import numpy as np
import pytest
import torch

from conftest import assert_close, load_golden, t
from test_oracle_cpu import CORR_CASES, expand_params

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import understanding_flow_robustness_amd as ufr
    from understanding_flow_robustness_amd import (alt_cuda_corr, channelnorm_cuda, resample2d_cuda,
                                                   spatial_correlation_sampler_backend as be)
    from understanding_flow_robustness_amd import _lib
    assert _lib.lib().ufr_device_count() >= 1
    return dict(be=be, alt=alt_cuda_corr, rs=resample2d_cuda, cn=channelnorm_cuda, ufr=ufr)


# ---------------------------------------------------------------------------- spatial correlation
@pytest.mark.parametrize("case", CORR_CASES)
def test_corr_matches_reference_golden(ops, case):
    z = load_golden(case)
    prm = expand_params(z["params"])
    a, b, go = t(z["input1"], DEV), t(z["input2"], DEV), t(z["grad_output"], DEV)
    out = ops["be"].forward(a, b, *prm)
    tol = dict(rtol=1e-9, atol_scale=1e-12) if a.dtype == torch.float64 else dict(rtol=1e-4, atol_scale=2e-6)
    assert_close(out, t(z["output"]), what=f"{case} forward", **tol)
    g1, g2 = ops["be"].backward(a, b, go, *prm)
    assert_close(g1, t(z["grad_input1"]), what=f"{case} grad1", **tol)
    assert_close(g2, t(z["grad_input2"]), what=f"{case} grad2", **tol)


def test_corr_index_outputs_bit_exact(ops):
    """north_star: index/argmax outputs bit-exact -- the best displacement per pixel."""
    z = load_golden("corr_flownetc_small_f32")
    prm = expand_params(z["params"])
    out = ops["be"].forward(t(z["input1"], DEV), t(z["input2"], DEV), *prm).cpu()
    ref = t(z["output"])
    B, ph, pw, H, W = ref.shape
    top2 = ref.view(B, ph * pw, H, W).topk(2, dim=1).values
    decided = (top2[:, 0] - top2[:, 1]) > 1e-4 * top2[:, 0].abs().clamp_min(1.0)  # ignore fp ties
    am_ref = ref.view(B, ph * pw, H, W).argmax(1)
    am_out = out.view(B, ph * pw, H, W).argmax(1)
    assert torch.equal(am_ref[decided], am_out[decided]) and decided.float().mean() > 0.99


def test_corr_full_size_digest_and_oracle(ops, oracle):
    """FlowNetC's real shape [1,256,48,160], patch 21, dilation_patch 2."""
    z = load_golden("corr_flownetc_full_digest")
    g = torch.Generator().manual_seed(int(z["seed"]))
    a = torch.randn(1, 256, 48, 160, generator=g)
    b = torch.randn(1, 256, 48, 160, generator=g)
    prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
    out = ops["be"].forward(a.to(DEV), b.to(DEV), *prm)
    go = torch.randn(out.shape, generator=g)
    g1, g2 = ops["be"].backward(a.to(DEV), b.to(DEV), go.to(DEV), *prm)
    for name, got, s, ab, idx, val in (("out", out, "out_sum", "out_abs", "out_idx", "out_val"),
                                       ("g1", g1, "g1_sum", "g1_abs", "g_idx", "g1_val"),
                                       ("g2", g2, "g2_sum", "g2_abs", "g_idx", "g2_val")):
        got = got.cpu()
        assert abs(float(got.double().abs().sum()) - float(z[ab])) <= 1e-6 * float(z[ab]), name
        assert abs(float(got.double().sum()) - float(z[s])) <= 1e-6 * float(z[ab]), name
        assert_close(got.flatten()[t(z[idx])], t(z[val]), rtol=1e-4, atol_scale=1e-5, what=name)
    ref = oracle.corr_forward(a, b, *prm)
    assert_close(out, ref, rtol=1e-4, atol_scale=2e-6, what="full forward vs oracle")
    r1, r2 = oracle.corr_backward(a, b, go, *prm)
    assert_close(g1, r1, rtol=1e-4, atol_scale=2e-6, what="full grad1 vs oracle")
    assert_close(g2, r2, rtol=1e-4, atol_scale=2e-6, what="full grad2 vs oracle")


@pytest.mark.parametrize("shape,patch,dp", [((8, 256, 48, 160), 21, 2), ((4, 196, 6, 20), 9, 1),
                                            ((2, 32, 96, 320), 9, 1), ((1, 256, 56, 128), 21, 2),
                                            ((3, 5, 7, 9), 9, 1), ((2, 64, 24, 81), 21, 2)])
def test_corr_fast_path_properties(ops, oracle, shape, patch, dp):
    """Size-independent checks at BASELINE sizes (batch 8, PWC pyramid levels, FlowNet2 width):
    bilinearity, adjointness <corr(a,b),g> = <a,g1> = <b,g2>, and a sampled oracle comparison."""
    g = torch.Generator().manual_seed(sum(shape) + patch)
    a = torch.randn(shape, generator=g).to(DEV)
    b = torch.randn(shape, generator=g).to(DEV)
    prm = (1, 1, patch, patch, 0, 0, 1, 1, dp, dp, 1, 1)
    out = ops["be"].forward(a, b, *prm)
    go = torch.randn(out.shape, generator=g).to(DEV)
    g1, g2 = ops["be"].backward(a, b, go, *prm)
    lhs = float((out.double() * go.double()).sum())
    scale = float((out.double() * go.double()).abs().sum())
    assert abs(lhs - float((a.double() * g1.double()).sum())) < 1e-5 * scale
    assert abs(lhs - float((b.double() * g2.double()).sum())) < 1e-5 * scale
    out2 = ops["be"].forward(2.0 * a, b, *prm)
    assert_close(out2, 2.0 * out, rtol=1e-6, atol_scale=1e-7, what="linearity")
    # zero-displacement channel is the plain channel dot product
    r = (patch - 1) // 2
    assert_close(out[:, r, r], (a * b).sum(1), rtol=1e-4, atol_scale=1e-5, what="centre tap")
    # oracle on the first sample (seconds on the CPU even for the big shapes)
    ref = oracle.corr_forward(a[:1].cpu(), b[:1].cpu(), *prm)
    assert_close(out[:1], ref, rtol=1e-4, atol_scale=2e-6, what="sample 0 vs oracle")
    r1, r2 = oracle.corr_backward(a[:1].cpu(), b[:1].cpu(), go[:1].cpu(), *prm)
    assert_close(g1[:1], r1, rtol=1e-4, atol_scale=2e-6, what="grad1 sample 0 vs oracle")
    assert_close(g2[:1], r2, rtol=1e-4, atol_scale=2e-6, what="grad2 sample 0 vs oracle")


def test_corr_kernels_are_bit_reproducible(ops):
    """No atomics on the correlation path: two launches give identical bits."""
    g = torch.Generator().manual_seed(4)
    a = torch.randn(2, 256, 48, 160, generator=g).to(DEV)
    b = torch.randn(2, 256, 48, 160, generator=g).to(DEV)
    prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
    o1, o2 = ops["be"].forward(a, b, *prm), ops["be"].forward(a, b, *prm)
    assert torch.equal(o1, o2)
    go = torch.randn(o1.shape, generator=g).to(DEV)
    x, y = ops["be"].backward(a, b, go, *prm), ops["be"].backward(a, b, go, *prm)
    assert torch.equal(x[0], y[0]) and torch.equal(x[1], y[1])


def test_corr_autograd_module_and_gradcheck(ops):
    """check.py / grad_check.py of the reference, on the device path (float64 finite differences)."""
    from understanding_flow_robustness_amd.spatial_correlation_sampler import SpatialCorrelationSampler
    g = torch.Generator().manual_seed(2)
    a = torch.randn(2, 2, 10, 10, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
    b = torch.randn(2, 2, 10, 10, dtype=torch.float64, generator=g).to(DEV).requires_grad_(True)
    sampler = SpatialCorrelationSampler(3, 3, 2, 1, 2, 2)
    assert torch.autograd.gradcheck(sampler, [a, b])


@pytest.mark.parametrize("case", ["corr_flownetc_small_f32", "corr_pwc_ragged_f32", "corr_rect_f32"])
def test_corr_float16_storage(ops, case):
    """The reference's CUDA op dispatches half as well (correlation_cuda_kernel.cu:262, :297).  Here: float16 storage,
    float32 sums -- compared with the float32 path (pinned to the reference by the tests above) on the same half-rounded
    inputs, at the resolution of one rounding to half; the golden's unrounded outputs bound the whole thing."""
    z = load_golden(case)
    prm = expand_params(z["params"])
    a, b, go = (t(z[k], DEV).half() for k in ("input1", "input2", "grad_output"))
    want = ops["be"].forward(a.float(), b.float(), *prm)
    w1, w2 = ops["be"].backward(a.float(), b.float(), go.float(), *prm)
    assert_close(want, t(z["output"]), rtol=3e-2, atol_scale=1e-2, what="rounded inputs move the output by half-epsilons only")
    out = ops["be"].forward(a, b, *prm)
    g1, g2 = ops["be"].backward(a, b, go, *prm)
    assert out.dtype == g1.dtype == g2.dtype == torch.float16
    for got, ref, what in ((out, want, "forward"), (g1, w1, "grad1"), (g2, w2, "grad2")):
        assert_close(got.float(), ref, rtol=1e-3, atol_scale=1e-3, what=f"{case} float16 {what}")     # one rounding to half


def test_corr_fused_epilogue(ops):
    g = torch.Generator().manual_seed(8)
    a = torch.randn(2, 16, 12, 20, generator=g).to(DEV)
    b = torch.randn(2, 16, 12, 20, generator=g).to(DEV)
    prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
    plain = ops["be"].forward(a, b, *prm)
    fused = ops["be"].forward(a, b, *prm, scale=1.0 / 16, slope=0.1)
    assert_close(fused, torch.nn.functional.leaky_relu(plain / 16, 0.1), rtol=1e-6, atol_scale=1e-7)


def test_corr_error_behaviour(ops):
    a = torch.zeros(1, 2, 4, 4, device=DEV)
    with pytest.raises(RuntimeError, match="contiguous"):
        ops["be"].forward(a.transpose(2, 3), a, 1, 1, 3, 3, 0, 0, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError):
        ops["be"].forward(a, torch.zeros(1, 3, 4, 4, device=DEV), 1, 1, 3, 3, 0, 0, 1, 1, 1, 1, 1, 1)
    with pytest.raises(RuntimeError, match="empty"):
        ops["be"].forward(a, a, 9, 9, 3, 3, 0, 0, 1, 1, 1, 1, 1, 1)


# ---------------------------------------------------------------------------- alt_cuda_corr
def _alt_inputs(B, H, W, C, g, spread=3.0, N=1):
    f1 = torch.randn(B, H, W, C, generator=g)
    f2 = torch.randn(B, H, W, C, generator=g)
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, N, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, N, H, W)
    coords = torch.stack([xs, ys], -1) + spread * torch.randn(B, N, H, W, 2, generator=g)
    return f1, f2, coords.contiguous()


def test_altcorr_matches_corrblock_golden(ops):
    """alt_cuda_corr values == CorrBlock values (reference's own identity, models/raft/corr.py)."""
    z = load_golden("raft_corrblock_lookup")
    f1, f2, coords = t(z["fmap1"], DEV), t(z["fmap2"], DEV), t(z["coords"], DEV)
    B, C, H, W = f1.shape
    outs, f2l = [], f2
    for i in range(4):
        c_i = (coords.permute(0, 2, 3, 1) / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
        (o,) = ops["alt"].forward(f1.permute(0, 2, 3, 1).contiguous(), f2l.permute(0, 2, 3, 1).contiguous(), c_i, 4)
        outs.append(o.squeeze(1))
        if i < 3:
            f2l = torch.nn.functional.avg_pool2d(f2l, 2, stride=2)
    alt = torch.stack(outs, dim=1).reshape(B, -1, H, W) / np.sqrt(C)
    assert_close(alt, t(z["output"]), rtol=1e-4, atol_scale=2e-6, what="alt_corr vs CorrBlock golden")


@pytest.mark.parametrize("B,H,W,C,r", [(1, 48, 160, 256, 4), (2, 13, 17, 64, 3), (1, 6, 20, 96, 4)])
def test_altcorr_vs_oracle(ops, oracle, B, H, W, C, r):
    g = torch.Generator().manual_seed(B * 100 + H)
    f1, f2, coords = _alt_inputs(B, H, W, C, g)
    (o,) = ops["alt"].forward(f1.to(DEV), f2.to(DEV), coords.to(DEV), r)
    (ref,) = oracle.altcorr_forward(f1, f2, coords, r)
    assert_close(o, ref, rtol=1e-4, atol_scale=2e-6, what="alt_corr forward")
    go = torch.randn(o.shape, generator=g)
    g1, g2, gc = ops["alt"].backward(f1.to(DEV), f2.to(DEV), coords.to(DEV), go.to(DEV), r)
    r1, r2, _ = oracle.altcorr_backward(f1, f2, coords, go, r)
    assert float(gc.abs().max()) == 0.0
    assert_close(g1, r1, rtol=1e-4, atol_scale=2e-6, what="alt_corr fmap1_grad")
    assert_close(g2, r2, rtol=1e-4, atol_scale=1e-5, what="alt_corr fmap2_grad (atomics)")


@pytest.mark.parametrize("case", ["pyramid_level", "wild_flow", "two_lookups", "ragged_channels", "smooth"])
def test_altcorr_backward_tiled_scatter_vs_oracle(ops, oracle, case):
    """The fmap2 adjoint accumulates in an LDS copy of the region a pixel tile's windows land in and falls
    back to global atomics outside it: pooled fmap2 (coords / 2), windows far outside the image, N = 2,
    a channel count that is not a multiple of the 32-channel slab, and a smooth flow (all-LDS path)."""
    g = torch.Generator().manual_seed(len(case))
    B, H, W, C, r, N, spread, lvl = 1, 24, 40, 64, 4, 1, 3.0, 1
    if case == "pyramid_level":
        lvl = 2
    elif case == "wild_flow":
        spread = 60.0
    elif case == "two_lookups":
        B, N = 2, 2
    elif case == "ragged_channels":
        C, r = 40, 3
    elif case == "smooth":
        spread = 0.3
    f1, _, coords = _alt_inputs(B, H, W, C, g, spread=spread, N=N)
    f2 = torch.randn(B, H // lvl, W // lvl, C, generator=g)
    coords = (coords / lvl).contiguous()
    (o,) = ops["alt"].forward(f1.to(DEV), f2.to(DEV), coords.to(DEV), r)
    (ref,) = oracle.altcorr_forward(f1, f2, coords, r)
    assert_close(o, ref, rtol=1e-4, atol_scale=2e-6, what=f"{case}: forward")
    go = torch.randn(o.shape, generator=g)
    g1, g2, _ = ops["alt"].backward(f1.to(DEV), f2.to(DEV), coords.to(DEV), go.to(DEV), r)
    r1, r2, _ = oracle.altcorr_backward(f1, f2, coords, go, r)
    assert_close(g1, r1, rtol=1e-4, atol_scale=2e-6, what=f"{case}: fmap1_grad")
    assert_close(g2, r2, rtol=1e-4, atol_scale=1e-5, what=f"{case}: fmap2_grad")


def test_altcorr_autograd_function(ops):
    from understanding_flow_robustness_amd.alt_cuda_corr import AltCorrFunction
    g = torch.Generator().manual_seed(77)
    f1, f2, coords = _alt_inputs(1, 6, 7, 32, g)
    f1, f2 = f1.to(DEV).requires_grad_(True), f2.to(DEV).requires_grad_(True)
    out = AltCorrFunction.apply(f1, f2, coords.to(DEV), 2)
    out.square().sum().backward()
    assert f1.grad is not None and f2.grad is not None and float(f1.grad.abs().sum()) > 0


@pytest.mark.parametrize("B,H,W,C,r,spread", [(1, 48, 160, 256, 4, 3.0), (2, 24, 40, 128, 3, 0.4), (1, 16, 40, 256, 4, 40.0),
                                                (1, 8, 21, 128, 4, 3.0)])
def test_altcorr_pyramid_on_the_matrix_cores_vs_oracle(oracle, B, H, W, C, r, spread):
    """AlternateCorrBlock.__call__ (corr.py:121-137) as ONE launch per direction on the fp32 matrix cores: all four levels
    of the forward, d / d fmap1 and d / d fmap2_l, against the CPU oracle level by level -- smooth and wild coordinate
    fields (the bounding box is data, not an assumption), a width that is not a multiple of the 16-pixel tile, and two
    lookups accumulating into one set of gradient buffers as RAFT's iterations do."""
    import math
    from understanding_flow_robustness_amd.flownets.raft_corr import AltCorrPyramidFunction, _SharedGrad
    g = torch.Generator().manual_seed(B * 1000 + H + int(spread))
    L = 4 if H % 8 == 0 else 2
    f1 = torch.randn(B, H, W, C, generator=g)
    f2s = [torch.randn(B, max(H >> i, 1), max(W >> i, 1), C, generator=g) for i in range(L)]
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    lookups = [(torch.cat([xs, ys], 1) + spread * torch.randn(B, 2, H, W, generator=g)).contiguous() for _ in range(2)]
    scale = 1.0 / math.sqrt(C)
    rd = 2 * r + 1
    f1d = f1.to(DEV).requires_grad_(True)
    f2d = [f.to(DEV).requires_grad_(True) for f in f2s]
    shared = _SharedGrad()
    outs = [AltCorrPyramidFunction.apply(f1d, c.to(DEV), r, scale, shared, *f2d) for c in lookups]
    gos = [torch.randn(o.shape, generator=g) for o in outs]
    want1, want2 = torch.zeros_like(f1), [torch.zeros_like(f) for f in f2s]
    for c, o, go in zip(lookups, outs, gos):
        for i in range(L):
            ci = (c.permute(0, 2, 3, 1) / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
            (ref,) = oracle.altcorr_forward(f1, f2s[i], ci, r)
            assert_close(o[:, i * rd * rd:(i + 1) * rd * rd], ref.squeeze(1) * scale, rtol=1e-4, atol_scale=2e-6, what=f"level {i} forward")
            r1, r2, _ = oracle.altcorr_backward(f1, f2s[i], ci, (go[:, i * rd * rd:(i + 1) * rd * rd] * scale).unsqueeze(1).contiguous(), r)
            want1 += r1
            want2[i] += r2
    torch.autograd.backward(outs, [go.to(DEV) for go in gos])
    assert_close(f1d.grad, want1, rtol=1e-4, atol_scale=5e-6, what="d / d fmap1 (two lookups, all levels)")
    for i in range(L):
        assert_close(f2d[i].grad, want2[i], rtol=1e-4, atol_scale=1e-5, what=f"d / d fmap2 level {i}")


@pytest.mark.parametrize("B,H,W,C,r,spread", [(1, 48, 160, 256, 4, 3.0), (2, 24, 40, 128, 3, 0.4), (1, 16, 40, 256, 4, 40.0),
                                                (1, 8, 21, 128, 4, 3.0), (2, 21, 37, 256, 3, 12.0), (1, 48, 160, 256, 4, 0.0)])
def test_altcorr_lookup_on_split_planes_vs_oracle(oracle, B, H, W, C, r, spread):
    """Round 6: the lookup on the bf16 matrix cores with float32 accuracy (csrc/raft_altcorr_planes.hip: three bf16 planes per
    operand, six products, an 8 x 16 pixel tile sharing the bounding box of its windows through LDS) against the CPU oracle level by
    level AND against round 3's exact-fp32 matrix-core kernel -- smooth, wild (multi-strip boxes, windows far outside the image),
    zero flow, sides that are not multiples of the 8 x 16 tile, two samples, both channel counts and radii; then coordinates that
    are not finite (the reference's kernel reads garbage there; this kernel returns zeros for such a pixel and leaves every other
    pixel untouched)."""
    import math
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd.flownets.raft_corr import AltCorrPlanes, AltCorrPyramidFunction
    g = torch.Generator().manual_seed(B * 1000 + H + int(spread))
    nl = 4 if H % 8 == 0 else 2
    f1 = torch.randn(B, H, W, C, generator=g)
    f2s = [torch.randn(B, max(H >> i, 1), max(W >> i, 1), C, generator=g) for i in range(nl)]
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    coords = (torch.cat([xs, ys], 1) + spread * torch.randn(B, 2, H, W, generator=g)).contiguous()
    scale, rd = 1.0 / math.sqrt(C), 2 * r + 1
    f1d, f2d = f1.to(DEV), [f.to(DEV) for f in f2s]
    assert AltCorrPlanes.served(f1d, f2d, r)
    planes = AltCorrPlanes(f1d, f2d)
    # the planes ARE the maps: p0 + p1 + p2 == v exactly, in the igemm's chunk-major layout
    p0 = planes.maps[0][0].float().sum(0).view(C // 32, B * H * W, 32).permute(1, 0, 2).reshape(B, H, W, C)
    assert torch.equal(p0, f1d)
    out = planes.forward(coords.to(DEV), r, scale)
    old = AltCorrPyramidFunction.apply(f1d, coords.to(DEV), r, scale, None, *f2d)
    for i in range(nl):
        ci = (coords.permute(0, 2, 3, 1) / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
        (ref,) = oracle.altcorr_forward(f1, f2s[i], ci, r)
        assert_close(out[:, i * rd * rd:(i + 1) * rd * rd], ref.squeeze(1) * scale, rtol=1e-4, atol_scale=2e-6, what=f"level {i} forward (split planes)")
    assert_close(out, old, rtol=1e-4, atol_scale=2e-6, what="split planes vs the exact-fp32 matrix-core kernel")
    bad = coords.clone().to(DEV)
    bad[0, 0, 0, 0], bad[0, 1, H - 1, W - 1], bad[-1, 0, H // 2, W // 2] = float("nan"), float("inf"), -float("inf")
    o2, old2 = planes.forward(bad, r, scale), AltCorrPyramidFunction.apply(f1d, bad, r, scale, None, *f2d)
    assert bool(torch.isfinite(o2).all())
    ok = torch.ones(B, 1, H, W, dtype=torch.bool, device=DEV)
    for bb, yy, xx in ((0, 0, 0), (0, H - 1, W - 1), (B - 1, H // 2, W // 2)):
        assert float(o2[bb, :, yy, xx].abs().max()) == 0.0
        ok[bb, 0, yy, xx] = False
    assert_close(o2 * ok, torch.where(ok, old2, torch.zeros_like(old2)), rtol=1e-4, atol_scale=2e-6, what="non-finite coordinates: the other pixels")


@pytest.mark.parametrize("B,H,W,C,r,spread", [(1, 48, 160, 256, 4, 3.0), (2, 24, 40, 128, 3, 0.4), (1, 16, 40, 256, 4, 40.0)])
def test_altcorr_dense_adjoint_vs_oracle(oracle, B, H, W, C, r, spread):
    """Round 6: the adjoints of ALL lookups of one AlternateCorrBlock at once (flownets/raft_corr.py::alt_dense_adjoint): every lookup's
    window adjoints are scattered into dense gradient volumes (`ufr_corr_lookup_backward`, one writer per pixel slice), the volumes of
    the lookups ADD, and d / d fmap1, d / d fmap2_l are 2 x levels igemm products at the end.  Against the CPU oracle's
    alt_cuda_corr backward (correlation_kernel.cu:122-256 restated), two lookups accumulated, level by level: the same gate as the
    on-the-fly adjoint kernels hold (test_altcorr_pyramid_on_the_matrix_cores_vs_oracle)."""
    import ctypes
    import math
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd.flownets.raft_corr import (_pyramid_struct, alt_dense_adjoint, alt_dense_adjoint_served,
                                                                      alt_dense_volumes)
    g = torch.Generator().manual_seed(B * 1000 + H + int(spread))
    nl = 4 if H % 8 == 0 else 2
    f1 = torch.randn(B, H, W, C, generator=g)
    f2s = [torch.randn(B, max(H >> i, 1), max(W >> i, 1), C, generator=g) for i in range(nl)]
    xs = torch.arange(W).float().view(1, 1, 1, W).expand(B, 1, H, W)
    ys = torch.arange(H).float().view(1, 1, H, 1).expand(B, 1, H, W)
    lookups = [(torch.cat([xs, ys], 1) + spread * torch.randn(B, 2, H, W, generator=g)).contiguous() for _ in range(2)]
    scale, rd = 1.0 / math.sqrt(C), 2 * r + 1
    gos = [torch.randn(B, nl * rd * rd, H, W, generator=g) for _ in lookups]
    want1, want2 = torch.zeros_like(f1), [torch.zeros_like(f) for f in f2s]
    for c, go in zip(lookups, gos):
        for i in range(nl):
            ci = (c.permute(0, 2, 3, 1) / 2 ** i).reshape(B, 1, H, W, 2).contiguous()
            r1, r2, _ = oracle.altcorr_backward(f1, f2s[i], ci, (go[:, i * rd * rd:(i + 1) * rd * rd] * scale).unsqueeze(1).contiguous(), r)
            want1 += r1
            want2[i] += r2
    f1d, f2d = f1.to(DEV), [f.to(DEV) for f in f2s]
    assert alt_dense_adjoint_served(f1d, f2d, r)
    vols = alt_dense_volumes(f1d, f2d)
    assert [tuple(v.shape) for v in vols] == [(B * H * W, 1, f.shape[1], f.shape[2]) for f in f2s]
    pyr = _pyramid_struct(vols, vols)
    for c, go in zip(lookups, gos):
        L.check(L.lib().ufr_corr_lookup_backward(ctypes.byref(pyr), L.ptr(c.to(DEV)), L.ptr(go.to(DEV)), B, H, W, r, L.stream()), "lookup adjoint")
    g1, g2 = alt_dense_adjoint(f1d, f2d, vols, scale)
    assert_close(g1, want1, rtol=1e-4, atol_scale=5e-6, what="d / d fmap1 (two lookups, all levels, dense adjoint)")
    for i in range(nl):
        assert_close(g2[i], want2[i], rtol=1e-4, atol_scale=1e-5, what=f"d / d fmap2 level {i} (dense adjoint)")
    # a second call through the cached launches, another batch element order: same bits
    g1b, g2b = alt_dense_adjoint(f1d, f2d, vols, scale)
    assert torch.equal(g1, g1b) and all(torch.equal(a, b) for a, b in zip(g2, g2b))


# ---------------------------------------------------------------------------- CorrBlock lookup
def test_lookup_matches_corrblock_golden_and_oracle(ops, oracle):
    from understanding_flow_robustness_amd.flownets.raft_corr import corr_lookup
    z = load_golden("raft_corrblock_lookup")
    f1, f2, coords = t(z["fmap1"], DEV), t(z["fmap2"], DEV), t(z["coords"], DEV)
    B, C, H, W = f1.shape
    corr = torch.matmul(f1.view(B, C, H * W).transpose(1, 2), f2.view(B, C, H * W)) / np.sqrt(C)
    pyr = [corr.view(B * H * W, 1, H, W)]
    for _ in range(3):
        pyr.append(torch.nn.functional.avg_pool2d(pyr[-1], 2, stride=2))
    pyr = [p.contiguous().requires_grad_(True) for p in pyr]
    out = corr_lookup(pyr, coords, 4)
    assert_close(out, t(z["output"]), rtol=1e-4, atol_scale=2e-6, what="lookup vs CorrBlock golden")
    ref = oracle.corr_lookup([p.detach().cpu() for p in pyr], coords.cpu(), 4)
    assert_close(out, ref, rtol=1e-5, atol_scale=1e-6, what="lookup vs oracle")
    # adjoint against torch autograd through grid_sample (the reference's own formulation)
    go = torch.randn(out.shape, generator=torch.Generator().manual_seed(1)).to(DEV)
    out.backward(go)
    pyr2 = [p.detach().clone().requires_grad_(True) for p in pyr]
    r = 4
    outs = []
    cperm = coords.permute(0, 2, 3, 1)
    for i, cv in enumerate(pyr2):
        d = torch.linspace(-r, r, 2 * r + 1, device=DEV)
        delta = torch.stack(torch.meshgrid(d, d, indexing="ij"), dim=-1)
        cl = cperm.reshape(B * H * W, 1, 1, 2) / 2 ** i + delta.view(1, 2 * r + 1, 2 * r + 1, 2)
        hh, ww = cv.shape[-2:]
        xg = 2 * cl[..., 0] / max(ww - 1, 1) - 1 if ww > 1 else cl[..., 0] * 0
        yg = 2 * cl[..., 1] / max(hh - 1, 1) - 1 if hh > 1 else cl[..., 1] * 0
        s = torch.nn.functional.grid_sample(cv, torch.stack([xg, yg], -1), align_corners=True)
        outs.append(s.view(B, H, W, -1))
    ref_out = torch.cat(outs, -1).permute(0, 3, 1, 2)
    ref_out.backward(go)
    for lvl in range(3):   # level 3 is 1x2 here: grid_sample's (W-1) normalisation degenerates
        assert_close(pyr[lvl].grad, pyr2[lvl].grad, rtol=1e-4, atol_scale=1e-5, what=f"lookup grad level {lvl}")


# ---------------------------------------------------------------------------- Resample2d / ChannelNorm
@pytest.mark.parametrize("B,C,H,W,bilinear,spread", [(2, 3, 17, 23, True, 4.0), (1, 3, 448, 1024, True, 4.0), (2, 2, 9, 11, False, 4.0),
                                                      (2, 3, 100, 260, True, 300.0),     # wild flow: the tiles' source boxes exceed LDS -> direct form
                                                      (1, 8, 70, 130, True, 1.5), (1, 3, 64, 96, True, 30.0)])   # mixed: some tiles staged, some not
@pytest.mark.parametrize("lds_mode", ["1", "2", "0", "3"])   # default (direct forward, owner-computes adjoint) / both LDS-staged / both direct / direct forward + LDS-privatised adjoint
def test_resample2d_vs_oracle(ops, oracle, monkeypatch, lds_mode, B, C, H, W, bilinear, spread):
    """csrc/resample2d_owner.hip (the adjoint without global atomics: one owner workgroup per tile of the image gradient) and
    csrc/warp_norm.hip (the LDS-staged forward / privatised image adjoint and the direct forms), against the C oracle's literal
    restatement of resample2d_kernel.cu:15-198.  The gradient buffers are handed over UNINITIALISED: every element is written."""
    monkeypatch.setenv("UFR_RESAMPLE_LDS", lds_mode)
    g = torch.Generator().manual_seed(H)
    img = torch.rand(B, C, H, W, generator=g)
    flow = spread * torch.randn(B, 2, H, W, generator=g)   # negative coordinates exercise int() vs floor()
    out = torch.empty(B, C, H, W, device=DEV)
    ops["rs"].forward(img.to(DEV), flow.to(DEV), out, 1, bilinear)
    ref = torch.empty(B, C, H, W)
    oracle.resample2d_forward(img, flow, ref, 1, bilinear)
    assert_close(out, ref, rtol=1e-6, atol_scale=1e-7, what="resample2d forward")
    go = torch.randn(B, C, H, W, generator=g)
    g1, g2 = torch.empty(B, C, H, W, device=DEV), torch.empty(B, 2, H, W, device=DEV)
    ops["rs"].backward(img.to(DEV), flow.to(DEV), go.to(DEV), g1, g2, 1, bilinear)
    r1, r2 = torch.empty(B, C, H, W), torch.empty(B, 2, H, W)
    oracle.resample2d_backward(img, flow, go, r1, r2, 1, bilinear)
    assert_close(g1, r1, rtol=1e-4, atol_scale=1e-5, what="resample2d grad image (atomics)")
    assert_close(g2, r2, rtol=1e-5, atol_scale=1e-6, what="resample2d grad flow")


def test_resample2d_module_identity_and_shift(ops):
    from understanding_flow_robustness_amd.resample2d_package.resample2d import Resample2d
    img = torch.rand(1, 3, 8, 12, device=DEV)
    assert torch.equal(Resample2d()(img, torch.zeros(1, 2, 8, 12, device=DEV)), img)
    flow = torch.zeros(1, 2, 8, 12, device=DEV)
    flow[:, 0] = 3.0
    xs = (torch.arange(12, device=DEV) + 3).clamp(0, 11)
    assert torch.equal(Resample2d()(img, flow), img[..., xs])


@pytest.mark.parametrize("B,C,H,W", [(2, 3, 17, 23), (1, 2, 448, 1024), (8, 3, 448, 1024)])
def test_channelnorm_vs_oracle(ops, oracle, B, C, H, W):
    g = torch.Generator().manual_seed(W)
    x = torch.randn(B, C, H, W, generator=g)
    out = torch.empty(B, 1, H, W, device=DEV)
    ops["cn"].forward(x.to(DEV), out, 2)
    ref = torch.empty(B, 1, H, W)
    oracle.channelnorm_forward(x, ref)
    assert_close(out, ref, rtol=1e-6, atol_scale=1e-7, what="channelnorm forward")
    go = torch.randn(B, 1, H, W, generator=g)
    gi = torch.empty(B, C, H, W, device=DEV)
    ops["cn"].backward(x.to(DEV), out, go.to(DEV), gi, 2)
    rg = torch.empty(B, C, H, W)
    oracle.channelnorm_backward(x, ref, go, rg)
    assert_close(gi, rg, rtol=1e-5, atol_scale=1e-6, what="channelnorm backward")


def test_channelnorm_module_autograd(ops):
    from understanding_flow_robustness_amd.channelnorm_package.channelnorm import ChannelNorm
    x = torch.randn(2, 3, 5, 7, device=DEV, requires_grad=True)
    y = ChannelNorm()(x)
    y.sum().backward()
    ref = x.detach() / (x.detach().pow(2).sum(1, keepdim=True).sqrt() + 1e-9)
    assert_close(x.grad, ref, rtol=1e-5, atol_scale=1e-6)


def test_resample2d_and_channelnorm_independent_pins_at_full_size(ops):
    """Parity of these two operators is UNPINNED BY THE REFERENCE (CUDA-only there, no vectors; the oracle is a literal
    restatement).  An independent pin for the forward at config C5's size, 448x1024: where every sample and its +1
    neighbour stay inside the image, Resample2d (resample2d_kernel.cu:15-72) is the bilinear gather that torch's
    grid_sample(align_corners=True) computes, and ChannelNorm (channelnorm_kernel.cu:18-60) is x.norm(dim=1); the adjoints
    against autograd of those spellings on the same samples."""
    B, C, H, W = 2, 3, 448, 1024
    g = torch.Generator().manual_seed(448)
    img = torch.rand(B, C, H, W, generator=g).to(DEV)
    flow = (6.0 * torch.rand(B, 2, H, W, generator=g) - 3.0).to(DEV)
    ys, xs = torch.meshgrid(torch.arange(H, device=DEV), torch.arange(W, device=DEV), indexing="ij")
    sx, sy = xs[None] + flow[:, 0], ys[None] + flow[:, 1]
    inside = ((sx >= 0) & (sx <= W - 2) & (sy >= 0) & (sy <= H - 2)).unsqueeze(1)          # no clamped corner, no int()-vs-floor() case
    assert float(inside.float().mean()) > 0.97
    # (grid_sample in float64: its normalised-coordinate round trip costs 1e-4 px in float32, the kernel samples at x + flow)
    img_r, flow_r = img.double().requires_grad_(True), flow.double().requires_grad_(True)
    grid = torch.stack((2 * (xs[None] + flow_r[:, 0]) / (W - 1) - 1, 2 * (ys[None] + flow_r[:, 1]) / (H - 1) - 1), -1)
    want = torch.nn.functional.grid_sample(img_r, grid, mode="bilinear", padding_mode="border", align_corners=True)
    out = torch.empty(B, C, H, W, device=DEV)
    ops["rs"].forward(img, flow, out, 1, True)
    err = float(((out - want.detach()) * inside).abs().max())
    # the reference forms x + flow in float32 (resample2d_kernel.cu:40-41): at x ~ 1000 the fractional weight carries up to
    # half an ulp(1024) = 6e-5 of error against exact arithmetic -- the operator's own rounding, reproduced bit for bit by
    # the kernel and the oracle alike; a wrong corner or weight is off by O(0.1)
    assert err <= 1.5e-4, f"Resample2d vs grid_sample on in-range samples: {err:.2e}"
    go = torch.randn(B, C, H, W, generator=g).to(DEV) * inside
    gi_ref, gf_ref = torch.autograd.grad(want, (img_r, flow_r), go.double())
    g1, g2 = torch.empty(B, C, H, W, device=DEV), torch.empty(B, 2, H, W, device=DEV)
    ops["rs"].backward(img, flow, go, g1, g2, 1, True)
    assert_close(g1, gi_ref, rtol=1e-3, atol_scale=2e-4, what="Resample2d image adjoint vs grid_sample's")
    # d / d flow is piecewise constant per bilinear cell: a sample within float32 rounding of an integer coordinate lies in the
    # neighbouring cell for one of the two evaluations
    frac = lambda v: (v - v.floor())
    smooth = inside & ((frac(sx) > 1e-3) & (frac(sx) < 1 - 1e-3) & (frac(sy) > 1e-3) & (frac(sy) < 1 - 1e-3)).unsqueeze(1)
    assert float(smooth.float().mean()) > 0.96
    assert_close(g2 * smooth, gf_ref * smooth, rtol=1e-3, atol_scale=2e-4, what="Resample2d flow adjoint vs grid_sample's")
    x = torch.randn(B, C, H, W, generator=g).to(DEV)
    outn = torch.empty(B, 1, H, W, device=DEV)
    ops["cn"].forward(x, outn, 2)
    xr = x.clone().requires_grad_(True)
    wantn = xr.norm(dim=1, keepdim=True)
    assert_close(outn, wantn.detach(), rtol=1e-6, atol_scale=1e-7, what="ChannelNorm vs x.norm(dim=1)")
    gon = torch.randn(B, 1, H, W, generator=g).to(DEV)
    (gn_ref,) = torch.autograd.grad(wantn, xr, gon)
    gin = torch.empty(B, C, H, W, device=DEV)
    ops["cn"].backward(x, outn, gon, gin, 2)
    assert_close(gin, gn_ref, rtol=1e-5, atol_scale=1e-6, what="ChannelNorm adjoint vs autograd of x.norm(dim=1)")

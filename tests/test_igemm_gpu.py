"""GPU suite: csrc/igemm.hip -- every convolution form of models/FlowNetC.py:22-50 (submodules.py:18-46 `conv`, :75-82
`deconv`) and their data gradients as ONE implicit-GEMM kernel on bf16 split planes, against torch's float32 operators
(float64 for the accuracy claim).  Tolerance: 1e-5 of max |result| -- the level at which MIOpen's own fp32 kernels differ
from float64 (profiles/r1_split_conv_accuracy.jsonl)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


VARIANT = {"v": 0}


class _Ig:
    """The igemm host module with `make_launch` pinned to one kernel form (csrc/igemm.hip: 2 = single-stage, 4 = 64 x 128
    tiles, 5 = pipelined, 6 = ping-pong, 7 = ping-pong with horizontal runs of taps staged once, 8 = the direct 3 x 3 form: a
    launch it does not cover -- other kernels, strides, more than 128 columns -- falls through to the plain forms)."""

    def __getattr__(self, name):
        from understanding_flow_robustness_amd import igemm
        if name == "make_launch":
            return lambda *a, **k: igemm.make_launch(*a, variant=VARIANT["v"], **k)
        return getattr(igemm, name)


@pytest.fixture(autouse=True, params=[2, 4, 5, 6, 7, 8],
                ids=["single-stage", "tile-64x128", "pipelined", "ping-pong", "ping-pong-tap-reuse", "direct-3x3"])
def _variant(request):
    VARIANT["v"] = request.param
    yield
    VARIANT["v"] = 0


def _mods():
    return _Ig()


def _close(got, want, what, tol=TOL):
    scale = float(want.abs().max())
    err = float((got.double() - want.double()).abs().max())
    assert err <= tol * scale, f"{what}: {err / scale:.2e} of max |result|"


def _rand(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(DEV)


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", [
    (2, 96, 160, 24, 40, 3, 1, 1),        # `conv` block, stride 1, ragged channel counts
    (3, 70, 130, 13, 21, 3, 2, 1),        # stride 2, odd sizes (M not a multiple of 128)
    (2, 256, 32, 12, 20, 1, 1, 0),        # conv_redir
    (1, 64, 128, 20, 36, 5, 2, 2),        # conv2 / conv3 shape (25 taps)
    (2, 160, 64, 24, 40, 3, 1, 1),        # PWC-Net's decoder: 64 outputs behind a long K (256 x 64 tiles in the tap-reuse form)
    (3, 117, 32, 13, 29, 3, 1, 1),        # ... 32 outputs, a row grid that ends inside a tile row, odd width
    (1, 6, 64, 40, 72, 3, 1, 1),          # FlowNetSD's conv0: 6 -> 64 at full resolution (one input chunk; the direct 3 x 3 form's case)
    (2, 11, 64, 17, 35, 3, 1, 1),         # FlowNetFusion's conv0, ragged tiles
    (1, 82, 16, 21, 70, 3, 1, 1),         # inter_conv0: 82 -> 16 (one column tile of 16)
])
def test_forward_convolution_with_bias_and_leaky(B, Cin, Cout, H, W, k, s, p):
    ig = _mods()
    x, w, b = _rand(B, Cin, H, W, seed=1), _rand(Cout, Cin, k, k, seed=2, scale=0.05), _rand(Cout, seed=3)
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    xin = ig.Planes(B, H, W, ig.pad32(Cin) // 32, DEV).load_nchw(x)
    # output lands at chunk 2 of a wider buffer (a torch.cat of the reference is a chunk offset here)
    out = ig.Planes(B, Ho, Wo, 2 + ig.pad32(Cout) // 32 + 1, DEV)
    out.t.fill_(7.0)
    wi = ig.conv_forward_weights(w, s, p)
    ig.make_launch(wi, xin, 0, (Ho, Wo), (Ho, Wo), out_planes=out, out_chunk0=2, bias=b)()
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), s, p), 0.1)
    _close(out.to_nchw(Cout, 2), want, "forward")
    # neighbours untouched, padding channels of the last chunk zero
    assert bool((out.t[:, :2] == 7.0).all()) and bool((out.t[:, 2 + ig.pad32(Cout) // 32:] == 7.0).all())
    if Cout % 32:
        assert bool((out.t[:, 2 + Cout // 32, :, Cout % 32:] == 0).all())


def test_round_trip_of_the_layout_passes_is_exact():
    ig = _mods()
    x = _rand(2, 70, 9, 13, seed=5) * 1e3
    pl = ig.Planes(2, 9, 13, 4, DEV).load_nchw(x, chunk0=1)
    assert torch.equal(pl.to_nchw(70, 1), x)                      # three bf16 planes hold a float32 exactly
    assert bool((pl.t[:, 3, :, 6:] == 0).all())
    y = ig.Planes(2, 9, 13, 3, DEV).load_nchw(x, scale=0.5, slope=0.1).to_nchw(70)
    assert torch.equal(y, F.leaky_relu(x * 0.5, 0.1))


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 96, 160, 12, 20), (1, 1026, 256, 6, 10)])
def test_deconv_forward_phases(B, Cin, Cout, H, W):
    """ConvTranspose2d(Cin, Cout, 4, 2, 1) + bias + LeakyReLU (submodules.py:75-82) as four phase GEMMs."""
    ig = _mods()
    x, w, b = _rand(B, Cin, H, W, seed=1), _rand(Cin, Cout, 4, 4, seed=2, scale=0.05), _rand(Cout, seed=3)
    xin = ig.Planes(B, H, W, ig.pad32(Cin) // 32, DEV).load_nchw(x)
    out = ig.Planes(B, 2 * H, 2 * W, ig.pad32(Cout) // 32, DEV)
    wi = ig.deconv_forward_weights(w, 1)
    ig.make_launch(wi, xin, 0, (H, W), (2 * H, 2 * W), out_planes=out, bias=b)()
    want = F.leaky_relu(F.conv_transpose2d(x.double(), w.double(), b.double(), 2, 1), 0.1)
    _close(out.to_nchw(Cout), want, "deconv forward")


@pytest.mark.parametrize("B,Cin,Cout,H,W,k,s,p", [(2, 96, 160, 24, 40, 3, 1, 1), (2, 70, 130, 24, 40, 3, 2, 1),
                                                   (1, 256, 32, 12, 20, 1, 1, 0), (2, 64, 224, 24, 40, 3, 1, 1),
                                                   (3, 32, 160, 13, 29, 3, 1, 1), (1, 6, 64, 40, 72, 3, 1, 1),
                                                   (2, 82, 16, 21, 70, 3, 1, 1)])
def test_data_gradient_with_addend_and_mask(B, Cin, Cout, H, W, k, s, p):
    """gx = (conv^T(gy) + addend) * LeakyReLU'(activation): planes and fp32 outputs of the gradient epilogue."""
    ig = _mods()
    Ho, Wo = (H + 2 * p - k) // s + 1, (W + 2 * p - k) // s + 1
    w, gy = _rand(Cout, Cin, k, k, seed=2, scale=0.05), _rand(B, Cout, Ho, Wo, seed=4)
    act, addend = _rand(B, Cin, H, W, seed=6), _rand(B, Cin, H, W, seed=7)
    x0 = torch.zeros(B, Cin, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
    (gx,) = torch.autograd.grad(F.conv2d(x0, w.double(), None, s, p), x0, gy.double())
    want = (gx + addend.double()) * torch.where(act > 0, 1.0, 0.1).double()
    gyp = ig.Planes(B, Ho, Wo, ig.pad32(Cout) // 32, DEV).load_nchw(gy)
    actp = ig.Planes(B, H, W, ig.pad32(Cin) // 32 + 1, DEV).load_nchw(act, chunk0=1)
    addg = ig.GradSum(B, H, W, ig.pad32(Cin) // 32, DEV)
    addg.t.copy_(ig.Planes(B, H, W, ig.pad32(Cin) // 32, DEV).load_nchw(addend).t.float().sum(0))
    outp = ig.Planes(B, H, W, ig.pad32(Cin) // 32, DEV)
    outf = ig.GradSum(B, H, W, ig.pad32(Cin) // 32, DEV)
    wi = ig.conv_backward_weights(w, s, p)
    rows = (Ho, Wo)
    ig.make_launch(wi, gyp, 0, rows, (H, W), out_planes=outp, out_f32=outf, add=addg, mask=actp, mask_chunk0=1)()
    _close(outp.to_nchw(Cin), want, "gradient planes")
    _close(outf.to_nchw(Cin), want, "gradient fp32")
    # the standalone finalize pass (several sources, last writer not a GEMM): g * leaky'(mask) -> planes
    from understanding_flow_robustness_amd import _lib as L
    fin = ig.Planes(B, H, W, ig.pad32(Cin) // 32, DEV)
    L.check(L.lib().ufr_grad_finalize(L.ptr(addg.t), 0, L.ptr(actp.t), 1, L.ptr(fin.t), fin.plane_stride, 0, fin.M,
                                      ig.pad32(Cin) // 32, 0.1, L.stream()))
    assert torch.equal(fin.to_nchw(Cin), addend * torch.where(act > 0, 1.0, 0.1))
    # and the fp32 -> NCHW pass with mask and scale (what the correlation's adjoint reads)
    got = outf.to_nchw(Cin, mask=actp, mask_chunk0=1, scale=0.25)
    _close(got, want * torch.where(act > 0, 1.0, 0.1).double() * 0.25, "gradient sum -> nchw")


@pytest.mark.parametrize("k,p,S", [(3, 1, 2), (3, 1, 4), (5, 2, 3), (5, 2, 9)])
def test_stride2_data_gradient_with_per_phase_slices(k, p, S):
    """A stride-2 data gradient's four phases reduce over unequal numbers of taps (1, 2, 2, 4 for k = 3; 4, 6, 6, 9 for
    k = 5): with split-K the longest phase gets S slices, the others proportionally fewer of the same length
    (csrc/igemm.hip: per_k / sk / zoff); the result equals the unsplit launch bit for bit when no phase is split, and the
    float64 gradient otherwise; two runs are bit-identical (fixed-order reduction)."""
    ig = _mods()
    B, Cin, Cout, H, W = 2, 64, 96, 24, 40
    Ho, Wo = H // 2, W // 2
    w, gy = _rand(Cout, Cin, k, k, seed=2, scale=0.05), _rand(B, Cout, Ho, Wo, seed=4)
    x0 = torch.zeros(B, Cin, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv2d(x0, w.double(), None, 2, p), x0, gy.double())
    gyp = ig.Planes(B, Ho, Wo, ig.pad32(Cout) // 32, DEV).load_nchw(gy)
    wi = ig.conv_backward_weights(w, 2, p)
    ws = torch.empty(len(wi.phases) * S * B * Ho * Wo * wi.Npad, dtype=torch.float32, device=DEV)
    outs = []
    for _ in range(2):
        out = ig.GradSum(B, H, W, ig.pad32(Cin) // 32, DEV)
        ig.make_launch(wi, gyp, 0, (Ho, Wo), (H, W), out_f32=out, splitk=S, ws=ws)()
        outs.append(out.to_nchw(Cin))
    assert torch.equal(outs[0], outs[1])
    _close(outs[0], want, f"stride-2 data gradient, k={k}, {S} slices")


def test_deconv_data_gradient():
    ig = _mods()
    B, Cin, Cout, H, W = 2, 96, 64, 12, 20
    w, gy = _rand(Cin, Cout, 4, 4, seed=2, scale=0.05), _rand(B, Cout, 2 * H, 2 * W, seed=4)
    x0 = torch.zeros(B, Cin, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv_transpose2d(x0, w.double(), None, 2, 1), x0, gy.double())
    gyp = ig.Planes(B, 2 * H, 2 * W, ig.pad32(Cout) // 32, DEV).load_nchw(gy)
    outf = ig.GradSum(B, H, W, ig.pad32(Cin) // 32, DEV)
    ig.make_launch(ig.deconv_backward_weights(w, 1), gyp, 0, (H, W), (H, W), out_f32=outf)()
    _close(outf.to_nchw(Cin), want, "deconv data gradient")


def test_split_k_equals_one_pass_and_is_reproducible():
    """conv6_1's shape class: 6x20 grid, K = 9*1024 -- too few tiles for 256 CUs without splitting the reduction."""
    ig = _mods()
    B, Cn, H, W = 2, 512, 6, 20
    x, w, b = _rand(B, Cn, H, W, seed=1), _rand(Cn, Cn, 3, 3, seed=2, scale=0.02), _rand(Cn, seed=3)
    xin = ig.Planes(B, H, W, Cn // 32, DEV).load_nchw(x)
    wi = ig.conv_forward_weights(w, 1, 1)
    one, many, again = (ig.Planes(B, H, W, Cn // 32, DEV) for _ in range(3))
    ig.make_launch(wi, xin, 0, (H, W), (H, W), out_planes=one, bias=b)()
    S = ig.splitk_for(B * H * W, wi.Npad, 9 * wi.KC)
    assert S >= 4
    ws = torch.empty(S * B * H * W * wi.Npad, device=DEV)
    for dst in (many, again):
        ig.make_launch(wi, xin, 0, (H, W), (H, W), out_planes=dst, bias=b, splitk=S, ws=ws)()
    want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1), 0.1)
    _close(one.to_nchw(Cn), want, "one pass")
    _close(many.to_nchw(Cn), want, "split-K")
    assert torch.equal(many.t, again.t)                         # fixed-order reduction: no atomics


@pytest.mark.parametrize("splitk", [1, 3])
def test_tail_columns_and_activation_with_addend(splitk):
    """PWC-Net's DenseNet forward push (pwc_engine.py): conv_2's launch carries conv_4's partial sum over the shared input
    chunks in its spare columns (raw fp32 `tail`), conv_4's own launch reduces over the rest and takes it back before its
    bias: both results equal the two plain convolutions."""
    ig = _mods()
    B, H, W = 2, 13, 29
    xa, xb = _rand(B, 64, H, W, seed=1), _rand(B, 96, H, W, seed=2)              # [front | shared] input chunks: 2 + 3
    w2, b2 = _rand(96, 96, 3, 3, seed=3, scale=0.05), _rand(96, seed=4)          # conv_2 reads the shared chunks only
    w4, b4 = _rand(32, 160, 3, 3, seed=5, scale=0.05), _rand(32, seed=6)         # conv_4 reads all five
    xin = ig.Planes(B, H, W, 5, DEV)
    xin.load_nchw(xa, 0)
    xin.load_nchw(xb, 2)
    out2, out4 = ig.Planes(B, H, W, 4, DEV), ig.Planes(B, H, W, 1, DEV)
    out2.t.fill_(7.0)
    part = ig.GradSum(B, H, W, 1, DEV)
    ws = torch.empty(splitk * B * H * W * 128, device=DEV)
    kw = dict(splitk=splitk, ws=ws) if splitk > 1 else {}
    wi2 = ig.conv_forward_weights(torch.cat((w2, w4[:, 64:]), 0), 1, 1)
    ig.make_launch(wi2, xin, 2, (H, W), (H, W), out_planes=out2, out_chunk0=0, bias=torch.cat((b2, torch.zeros(32, device=DEV))),
                   tail=part, tail_n0=96, **kw)()
    assert bool((out2.t[:, 3] == 7.0).all())                                     # the tail columns never reach the planes
    _close(out2.to_nchw(96, 0), F.leaky_relu(F.conv2d(xb.double(), w2.double(), b2.double(), 1, 1), 0.1), "conv_2")
    _close(part.to_nchw(32, 0, slope=1.0), F.conv2d(xb.double(), w4[:, 64:].double(), None, 1, 1), "conv_4's partial")
    wi4 = ig.conv_forward_weights(w4[:, :64].contiguous(), 1, 1)
    ig.make_launch(wi4, xin, 0, (H, W), (H, W), out_planes=out4, bias=b4, add=part, add_chunk0=0, **kw)()
    want = F.leaky_relu(F.conv2d(torch.cat((xa, xb), 1).double(), w4.double(), b4.double(), 1, 1), 0.1)
    _close(out4.to_nchw(32, 0), want, "conv_4")


def test_column_band_rows_and_input_band():
    """Rows restricted to a per-sample column band (origins in device memory): only the band's columns are written, with
    the full-frame result; an input band makes columns outside it read as zero."""
    ig = _mods()
    B, Cin, Cout, H, W, bw = 3, 64, 96, 10, 48, 16
    x, w, b = _rand(B, Cin, H, W, seed=1), _rand(Cout, Cin, 3, 3, seed=2, scale=0.05), _rand(Cout, seed=3)
    origins = torch.tensor([[0, 8 * 0, 0, 0], [0, 8 * 32, 0, 0], [0, 8 * 17, 0, 0]], dtype=torch.int32, device=DEV)  # pixels, cell = 8
    xin = ig.Planes(B, H, W, 2, DEV).load_nchw(x)
    out = ig.Planes(B, H, W, 3, DEV)
    out.t.fill_(7.0)
    wi = ig.conv_forward_weights(w, 1, 1)
    band = (origins[:, 1], 4, 8)               # tensor view starting at column 1, element stride 4, pixels -> cells
    ig.make_launch(wi, xin, 0, (H, bw), (H, W), out_planes=out, bias=b, row_band=band)()
    full = F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.1)
    got = out.to_nchw(Cout)
    for bi, x0 in enumerate((0, 32, 17)):
        _close(got[bi, :, :, x0:x0 + bw], full[bi, :, :, x0:x0 + bw].double(), f"band of sample {bi}")
        outside = torch.ones(W, dtype=torch.bool, device=DEV)
        outside[x0:x0 + bw] = False
        assert bool((out.t.float().sum(0).view(3, B, H, W, 32)[:, bi][:, :, outside] == 21.0).all())
    # input band: the same rows, but input columns outside [x0, x0 + bw) count as zero
    out2 = ig.Planes(B, H, W, 3, DEV)
    ig.make_launch(wi, xin, 0, (H, bw), (H, W), out_planes=out2, bias=b, row_band=band, in_band=band + (bw,))()
    got2 = out2.to_nchw(Cout)
    for bi, x0 in enumerate((0, 32, 17)):
        xm = torch.zeros_like(x[bi:bi + 1])
        xm[..., x0:x0 + bw] = x[bi:bi + 1, ..., x0:x0 + bw]
        ref = F.leaky_relu(F.conv2d(xm, w, b, 1, 1), 0.1)
        _close(got2[bi:bi + 1, :, :, x0:x0 + bw], ref[..., x0:x0 + bw].double(), f"input band of sample {bi}")


def test_accuracy_matches_fp32_against_float64():
    """The six-product split is an implementation of the float32 product: its error against float64 is at the level of
    torch's own float32 convolution on this device (conv3_1's reduction length, K = 9*480)."""
    ig = _mods()
    B, Cin, Cout, H, W = 1, 473, 256, 24, 40
    x, w, b = _rand(B, Cin, H, W, seed=1), _rand(Cout, Cin, 3, 3, seed=2, scale=0.03), _rand(Cout, seed=3)
    xin = ig.Planes(B, H, W, 15, DEV).load_nchw(x)
    out = ig.Planes(B, H, W, 8, DEV)
    ig.make_launch(ig.conv_forward_weights(w, 1, 1), xin, 0, (H, W), (H, W), out_planes=out, bias=b)()
    truth = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1), 0.1)
    fp32 = F.leaky_relu(F.conv2d(x, w, b, 1, 1), 0.1)
    scale = float(truth.abs().max())
    e_mine = float((out.to_nchw(Cout).double() - truth).abs().max()) / scale
    e_fp32 = float((fp32.double() - truth).abs().max()) / scale
    print(f"six-product split {e_mine:.2e}, torch fp32 {e_fp32:.2e} of max |out|")
    assert e_mine <= max(3 * e_fp32, 3e-6)


@pytest.mark.parametrize("case", ["conv6_1 forward", "stride-2 data gradient", "gru gates at 48x160"])
def test_fused_split_k_reduction_is_bit_identical_to_the_reduce_launch(case):
    """Round 6: with `tickets` the last workgroup to arrive at a tile adds the tile's slabs in ascending slice
    order and runs the epilogue -- no second launch.  Same slabs, same order, same epilogue code (`sum_slabs`, `epilogue_store8`):
    every bit of the result equals the slab kernel + `igemm_reduce_kernel` form, on every replay (the last arriver leaves the counters
    at zero), for a forward launch with bias + LeakyReLU -> planes, a four-phase stride-2 data gradient with unequal slices per
    phase + addend + mask -> fp32, and the GRU's 1 x 5 gate convolution on RAFT's 48 x 160 grid (30 - 120 tiles of 4 - 8 slices)."""
    ig = _mods()
    if case == "conv6_1 forward":
        B, Cn, H, W = 2, 512, 6, 20
        x, w, b = _rand(B, Cn, H, W, seed=1), _rand(Cn, Cn, 3, 3, seed=2, scale=0.02), _rand(Cn, seed=3)
        xin = ig.Planes(B, H, W, Cn // 32, DEV).load_nchw(x)
        wi = ig.conv_forward_weights(w, 1, 1)
        S = max(4, ig.splitk_for(B * H * W, wi.Npad, 9 * wi.KC))
        mk = lambda fuse, ws: (lambda out: (ig.make_launch(wi, xin, 0, (H, W), (H, W), out_planes=out, bias=b, splitk=S, ws=ws, fuse_reduce=fuse), out))(
            ig.Planes(B, H, W, Cn // 32, DEV))
        read = lambda out: out.t.clone()
        rows, nph = B * H * W, 1
        want = F.leaky_relu(F.conv2d(x.double(), w.double(), b.double(), 1, 1), 0.1)
        check = lambda out: _close(out.to_nchw(Cn), want, case)
    elif case == "stride-2 data gradient":
        B, Cin, Cout, H, W, S = 2, 70, 130, 24, 40, 3
        w, gy = _rand(Cout, Cin, 3, 3, seed=2, scale=0.05), _rand(B, Cout, H // 2, W // 2, seed=4)
        act, addv = _rand(B, Cin, H, W, seed=5), _rand(B, Cin, H, W, seed=6)
        gyp = ig.Planes(B, H // 2, W // 2, ig.pad32(Cout) // 32, DEV).load_nchw(gy)
        maskp = ig.Planes(B, H, W, ig.pad32(Cin) // 32, DEV).load_nchw(act)
        wi = ig.conv_backward_weights(w, 2, 1)
        nch = ig.pad32(Cin) // 32

        addcm = ig.Planes(B, H, W, nch, DEV).load_nchw(addv).t.float().sum(0)

        def mk(fuse, ws):          # the addend IS the output (an in-place accumulation, as RAFT's encoder and GRU adjoints run it):
            out = ig.GradSum(B, H, W, nch, DEV)     # a reduction that ran twice would add it twice
            return ig.make_launch(wi, gyp, 0, (H // 2, W // 2), (H, W), out_f32=out, add=out, mask=maskp, splitk=S, ws=ws, fuse_reduce=fuse), out
        reset = lambda out: out.t.copy_(addcm.view_as(out.t))
        read = lambda out: out.t.clone()
        rows, nph = B * (H // 2) * (W // 2), 4
        x0 = torch.zeros(B, Cin, H, W, device=DEV, dtype=torch.float64, requires_grad=True)
        (gx,) = torch.autograd.grad(F.conv2d(x0, w.double(), None, 2, 1), x0, gy.double())
        want = (gx + addv.double()) * torch.where(act > 0, 1.0, 0.1).double()
        check = lambda out: _close(out.to_nchw(Cin), want, case)
    else:
        B, Cn, H, W, S = 1, 256, 48, 160, 5
        x, w, b = _rand(B, Cn, H, W, seed=1), _rand(256, Cn, 1, 5, seed=2, scale=0.03), _rand(256, seed=3)
        xin = ig.Planes(B, H, W, Cn // 32, DEV).load_nchw(x)
        wi = ig.conv_forward_weights(w, 1, (0, 2))
        mk = lambda fuse, ws: (lambda out: (ig.make_launch(wi, xin, 0, (H, W), (H, W), out_f32=out, bias=b, slope=1.0, splitk=S, ws=ws, fuse_reduce=fuse), out))(
            ig.GradSum(B, H, W, 8, DEV))
        read = lambda out: out.t.clone()
        rows, nph = B * H * W, 1
        want = F.conv2d(x.double(), w.double(), b.double(), 1, (0, 2))
        check = lambda out: _close(out.to_nchw(256, slope=1.0), want, case)
    ws_a = torch.empty(nph * S * rows * wi.Npad, device=DEV)
    ws_b = torch.empty_like(ws_a)
    two_launches, out_two = mk(False, ws_a)
    fused, out_fused = mk(True, ws_b)
    assert two_launches.desc.tickets is None and fused.desc.tickets
    if case != "stride-2 data gradient":
        reset = lambda out: out.t.zero_()
    reset(out_two)
    two_launches()
    ref = read(out_two)
    check(out_two)
    tickets = next(t for t in fused._keep if torch.is_tensor(t) and t.dtype == torch.int32)
    for replay in range(3):
        reset(out_fused)
        ws_b.fill_(float("nan"))                                  # nothing stale can pass for a slab
        fused()
        assert torch.equal(read(out_fused), ref), f"{case}: replay {replay} differs from the slab kernel + reduce launch"
        assert int(tickets.abs().sum()) == 0, "the last arrivers left their counters non-zero"


@pytest.mark.parametrize("products", [3, 1])
def test_reduced_products_are_exactly_the_leading_bf16_products(products):
    """RAFT's opt-in reduced precision (`products` 1 / 3, ABI 9; models/utils_model.py:51, models/raft/raft.py:140,168,195 run the
    reference's convolutions under fp16 autocast): ONE product = the convolution of the bf16-ROUNDED operands with float32
    accumulation (what a bfloat16 autocast computes); THREE = a0b0 + a0b1 + a1b0 = (a0 + a1)(b0 + b1) - a1 b1.  Both pinned against
    float64 evaluations of exactly those expressions (so the form cannot silently drop or add a product), and their distance from
    the float32 convolution printed: ~2^-9 and ~2^-17 of the result's scale."""
    ig = _mods()
    B, Cin, Cout, H, W = 1, 256, 128, 24, 40
    x, w, b = _rand(B, Cin, H, W, seed=1), _rand(Cout, Cin, 3, 3, seed=2, scale=0.03), _rand(Cout, seed=3)
    xin = ig.Planes(B, H, W, Cin // 32, DEV).load_nchw(x)
    out = ig.GradSum(B, H, W, Cout // 32, DEV)
    S = 2
    ws = torch.empty(S * B * H * W * 128, device=DEV)
    for kw in ({}, dict(splitk=S, ws=ws)):
        ig.make_launch(ig.conv_forward_weights(w, 1, 1), xin, 0, (H, W), (H, W), out_f32=out, bias=b, slope=1.0, products=products, **kw)()
        got = out.to_nchw(Cout, slope=1.0).double()
        conv = lambda a, c: F.conv2d(a.double(), c.double(), None, 1, 1)
        x0, w0 = x.bfloat16().float(), w.bfloat16().float()
        x1, w1 = (x - x0).bfloat16().float(), (w - w0).bfloat16().float()
        want = conv(x0, w0) if products == 1 else conv(x0 + x1, w0 + w1) - conv(x1, w1)
        want = want + b.double().view(1, -1, 1, 1)
        full = F.conv2d(x.double(), w.double(), b.double(), 1, 1)
        scale = float(full.abs().max())
        e_form, e_full = float((got - want).abs().max()) / scale, float((got - full).abs().max()) / scale
        print(f"{products} product(s): {e_form:.2e} from its own expression, {e_full:.2e} from the float32 convolution (of max |out|)")
        assert e_form <= 3e-6
        assert e_full <= (2e-2 if products == 1 else 1e-4)

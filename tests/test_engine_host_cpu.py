"""CPU suite for the host logic of the native FlowNetC engine (igemm.py / flownetc_engine.py): the weight packings and
tap geometries that turn conv1 (packed planes), predict_flow and the deconvolutions' flow columns into GEMM launches are
checked against torch convolutions with a plain-torch emulation of what the HIP kernels compute -- no GPU needed."""
import torch
import torch.nn.functional as F

from understanding_flow_robustness_amd import igemm as ig


def _planes_value(wi: ig.WeightImage, phase: int = 0) -> torch.Tensor:
    """[taps*KC][Npad][32] float32 image of a phase: the three bf16 planes added back (exact)."""
    taps = len(wi.phases[phase][2])
    n = taps * wi.KC * wi.Npad * 32
    off = wi.offsets[phase]
    return wi.planes.float().sum(0)[off:off + n].view(taps * wi.KC, wi.Npad, 32)


def _pack_frames(x: torch.Tensor) -> torch.Tensor:
    """Emulation of csrc/plane_layout.hip conv1_pack_kernel without the mean: [N,3,H,W] -> [N, H/2+3, W/2+2, 32]."""
    N, _, H, W = x.shape
    Hh, Wh = H // 2, W // 2
    out = torch.zeros(N, Hh + 3, Wh + 2, 32, dtype=x.dtype)
    for j in range(2):
        for c in range(3):
            for p in range(2):
                for q in range(2):
                    ch = j * 12 + (c * 2 + p) * 2 + q
                    src = x[:, c, p::2, q::2]                              # [N, Hh, Wh] = frame[c, 2 yh + p, 2 xh + q]
                    # packed pixel (yp, xp) holds half-grid pixel (yp - 2, xp - 2 + j)
                    x_lo = 2 - j
                    out[:, 2:2 + Hh, x_lo:x_lo + Wh, ch] = src[:, :, :Wh + 2 - x_lo] if x_lo + Wh <= Wh + 2 else src
    return out


def test_conv1_packed_launch_is_the_7x7_stride2_convolution():
    """conv1_packed_weights + the packed planes: sum over the 8 taps (a, 2 b2) of one 32-channel chunk == Conv2d(3, N, 7, 2, 3)."""
    g = torch.Generator().manual_seed(0)
    w = torch.randn(16, 3, 7, 7, generator=g, dtype=torch.float64)
    x = torch.randn(2, 3, 12, 20, generator=g, dtype=torch.float64)
    wi = ig.conv1_packed_weights(w.float())
    taps = wi.phases[0][2]
    assert len(taps) == 8 and wi.KC == 1 and wi.geometry == dict(in_s=1, out_s=1)
    wimg = _planes_value(wi)                                                # [8][Npad][32]
    P = _pack_frames(x)
    Hh, Wh = 6, 10
    out = torch.zeros(2, Hh, Wh, wi.Npad, dtype=torch.float64)
    for t, (dy, dx) in enumerate(taps):
        out += torch.einsum("nyxc,oc->nyxo", P[:, dy:dy + Hh, dx:dx + Wh].double(), wimg[t].double())
    want = F.conv2d(x, w.float().double(), None, 2, 3)
    assert torch.allclose(out[..., :16].permute(0, 3, 1, 2), want, atol=1e-9)
    assert float(out[..., 16:].abs().max()) == 0.0


def test_conv1_packed_backward_launch_and_unpacking_are_the_data_gradient():
    """conv1_packed_backward_weights (taps (-a, -2 b2) over the packed grid) + the unpacking (a frame pixel sits in two packed
    pixels) == the data gradient of Conv2d(3, N, 7, 2, 3)."""
    g = torch.Generator().manual_seed(1)
    N = 8
    w = torch.randn(N, 3, 7, 7, generator=g)
    H, W, Hh, Wh = 12, 20, 6, 10
    gy = torch.randn(2, N, Hh, Wh, generator=g, dtype=torch.float64)
    wi = ig.conv1_packed_backward_weights(w)
    wimg = _planes_value(wi).double()                                       # [8 taps][Npad=64][32 (o)]
    Hp, Wp = Hh + 3, Wh + 2
    gP = torch.zeros(2, Hp, Wp, wi.Npad, dtype=torch.float64)
    gyh = gy.permute(0, 2, 3, 1)                                            # [n, Y, X, o]
    for t, (dy, dx) in enumerate(wi.phases[0][2]):
        # gP(yp, xp) += W_t^T gy(yp + dy, xp + dx), zero outside the (Hh, Wh) grid
        for yp in range(Hp):
            Y = yp + dy
            if not 0 <= Y < Hh:
                continue
            xs = [xp for xp in range(Wp) if 0 <= xp + dx < Wh]
            gP[:, yp, xs] += torch.einsum("nxo,co->nxc", gyh[:, Y, [xp + dx for xp in xs], :N], wimg[t][:, :N])
    gx = torch.zeros(2, 3, H, W, dtype=torch.float64)
    for c in range(3):
        for y in range(H):
            for x_ in range(W):
                r = (c * 2 + (y & 1)) * 2 + (x_ & 1)
                gx[:, c, y, x_] = gP[:, (y >> 1) + 2, (x_ >> 1) + 2, r] + gP[:, (y >> 1) + 2, (x_ >> 1) + 1, 12 + r]
    x0 = torch.zeros(2, 3, H, W, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv2d(x0, w.double(), None, 2, 3), x0, gy)
    assert torch.allclose(gx, want, atol=1e-9)


def test_predict_flow_and_deconv_tail_packings():
    """_pack_flow_head_mfma: T[p, 2k + o] = sum_c x[p, c] w[o, c, k] gathered over the 9 taps == Conv2d(C, 2, 3, 1, 1);
    _pack_flow_tail_mfma: the same with 4 x 4 stride-2 taps == the data gradient of ConvTranspose2d(., ., 4, 2, 1) with
    respect to two input channels (csrc/engine_small.hip flow_head_planes_fwd_mfma<0 / 1>)."""
    from understanding_flow_robustness_amd.flownetc_engine import _pack_flow_head_mfma, _pack_flow_tail_mfma
    g = torch.Generator().manual_seed(2)
    C, H, W = 70, 6, 9
    w = torch.randn(2, C, 3, 3, generator=g)
    x = torch.randn(1, C, H, W, generator=g, dtype=torch.float64)
    wm = _pack_flow_head_mfma(w).float().sum(1).double()                    # [chunks][32 n][32 c]
    chunks = wm.shape[0]
    xp = torch.zeros(chunks * 32, H + 2, W + 2, dtype=torch.float64)
    xp[:C, 1:-1, 1:-1] = x[0]
    T = torch.einsum("kcyx,knc->nyx", xp.view(chunks, 32, H + 2, W + 2), wm)   # [32 n][H+2][W+2]
    out = torch.zeros(2, H, W, dtype=torch.float64)
    for k in range(9):
        for o in range(2):
            out[o] += T[2 * k + o, k // 3:k // 3 + H, k % 3:k % 3 + W]
    assert torch.allclose(out, F.conv2d(x, w.double(), None, 1, 1)[0], atol=1e-9)
    assert float(T[18:].abs().max()) == 0.0
    # deconv tail: gy on the fine grid [Cout, 2H, 2W], weights of the two flow input channels [2, Cout, 4, 4]
    Cout = 40
    w2 = torch.randn(2, Cout, 4, 4, generator=g)
    gy = torch.randn(1, Cout, 2 * H, 2 * W, generator=g, dtype=torch.float64)
    wt = _pack_flow_tail_mfma(w2).float().sum(1).double()
    ch2 = wt.shape[0]
    gp = torch.zeros(ch2 * 32, 2 * H + 2, 2 * W + 2, dtype=torch.float64)    # fine grid with one pixel of zero halo
    gp[:Cout, 1:-1, 1:-1] = gy[0]
    T2 = torch.einsum("kcyx,knc->nyx", gp.view(ch2, 32, 2 * H + 2, 2 * W + 2), wt)
    got = torch.zeros(2, H, W, dtype=torch.float64)
    for k in range(16):
        ky, kx = k // 4, k % 4
        for o in range(2):
            got[o] += T2[2 * k + o, ky:ky + 2 * H:2, kx:kx + 2 * W:2]          # fine pixel (2y - 1 + ky, 2x - 1 + kx) + halo offset 1
    x0 = torch.zeros(1, 2, H, W, dtype=torch.float64, requires_grad=True)
    (want,) = torch.autograd.grad(F.conv_transpose2d(x0, w2.double(), None, 2, 1), x0, gy)
    assert torch.allclose(got, want[0], atol=1e-9)


def test_split_k_cost_model_keeps_uniform_launches_and_balances_unequal_phases():
    """splitk_for: launches with equal phases keep the simple rule (fill ~3 workgroups per CU, >= 16 K tiles per slice);
    a stride-2 data gradient (K tiles 16, 32, 32, 64) is split so that no workgroup is longer than half the longest phase."""
    assert ig.splitk_for(61440, 256, 135) == 1                       # conv3_1 forward: 960 workgroups already
    assert ig.splitk_for(960, 1024, 144) == 8                        # conv6 forward: 64 tiles
    assert ig.splitk_for(3840, 512, 144) == 4                        # conv5 forward
    pk = [16, 32, 32, 64]
    s = ig.splitk_for(7296, 256, 64, 4, phase_ktiles=pk)             # conv4 backward on the band
    assert s in (2, 4) and -(-max(pk) // s) <= 32
    assert ig.splitk_for(7296, 256, 64, 4, phase_ktiles=[64, 64, 64, 64]) == ig.splitk_for(7296, 256, 64, 4)


def test_conv1_direct_weight_image_is_the_7x7_stride2_convolution():
    """igemm.conv1_direct_weights: the image `ufr_conv1_direct` keeps in registers -- K tile = kernel row ky, K group g = the
    input pixel pair (2g, 2g + 1) x (3 channels + 1 zero) -- against Conv2d(3, 64, 7, 2, 3) through a plain-torch emulation of
    the kernel's im2col (the region starts at input column 2 X0 - 3, so output pixel x reads LDS pixels 2 x .. 2 x + 7)."""
    g = torch.Generator().manual_seed(2)
    w = torch.randn(64, 3, 7, 7, generator=g)
    x = torch.randn(2, 3, 10, 14, generator=g, dtype=torch.float64)
    img = ig.conv1_direct_weights(w)
    assert img.shape == (3, 7, 64, 32) and img.dtype == torch.bfloat16
    wv = img.float().sum(0).double()                                       # [7 ky][64 n][32 k], exact
    assert torch.equal(wv.float().sum(0).sum(0)[torch.tensor([3, 7, 11, 15, 19, 23, 27, 28, 29, 30, 31])], torch.zeros(11))   # channel 3 and kx = 7 carry zeros
    xp = F.pad(x, (3, 4, 3, 3))                                            # column 2 x0 - 3 ... one extra column for kx = 7
    Ho, Wo = 5, 7
    out = torch.zeros(2, 64, Ho, Wo, dtype=torch.float64)
    for ky in range(7):
        for gg in range(4):
            for j in range(2):
                for c in range(3):
                    k = gg * 8 + j * 4 + c
                    patch = xp[:, c, ky:ky + 2 * Ho:2, 2 * gg + j:2 * gg + j + 2 * Wo:2]      # input (2 y - 3 + ky, 2 x - 3 + 2 g + j)
                    out += wv[ky, :, k].view(1, 64, 1, 1) * patch.unsqueeze(1)
    want = F.conv2d(x, w.double(), None, 2, 3)
    assert torch.allclose(out, want, atol=1e-9)


def test_planes_as_weights_describe_the_all_pairs_product():
    """igemm.planes_as_weights: an activation's planes [3][chunks][M][32] read as a one-tap weight image [KC][Npad = M][32]
    (RAFT's all-pairs correlation, models/raft/corr.py:57-64): geometry only, no launch."""
    import pytest
    p = ig.Planes(1, 8, 16, 4, "cpu")                                      # M = 128 pixels, 128 channels
    wi = ig.planes_as_weights(p)
    assert (wi.N, wi.Npad, wi.KC, wi.phases, wi.offsets) == (128, 128, 4, [(0, 0, [(0, 0)])], [0])
    assert wi.planes.shape == (3, p.plane_stride) and wi.planes.data_ptr() == p.t.data_ptr() and wi.k_order == 1
    assert wi.flops(128) == 2.0 * 128 * 128 * 128
    with pytest.raises(ValueError):
        ig.planes_as_weights(ig.Planes(1, 5, 9, 4, "cpu"))                 # 45 pixels: not a multiple of 64


def test_engine_cache_is_lru_and_counts_evictions():
    """_lib.EngineCache: a hit makes an entry the most recently USED one; the seventh configuration evicts the least
    recently used, counts it and warns once per cache."""
    import warnings

    from understanding_flow_robustness_amd import _lib as L
    c = L.EngineCache()
    before = L.EngineCache.evictions
    for k in range(6):
        c[k] = f"engine{k}"
    assert c.get(0) == "engine0"                                           # 0 is now the most recently used
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        c[6] = "engine6"
        c[7] = "engine7"
    assert 0 in c and 1 not in c and 2 not in c and len(c) == 6
    assert L.EngineCache.evictions == before + 2 and len([m for m in w if "engine configurations" in str(m.message)]) == 1
    assert c.get("absent", "dflt") == "dflt"


def test_engine_refusal_names_the_reason():
    """_lib.engine_refusal / engine_gate: why a forward leaves the hand-written engines."""
    from understanding_flow_robustness_amd import _lib as L
    m = torch.nn.Conv2d(3, 4, 3).eval()
    assert L.engine_refusal(m, torch.zeros(1, 3, 64, 64), 64) == "not a HIP float32 tensor"

    class FakeHip:                                                         # a stand-in with the fields the gate reads
        is_cuda, dtype = True, torch.float32

        def __init__(self, h, w):
            self.shape = (1, 3, h, w)
    assert "require gradients" in L.engine_refusal(m, FakeHip(64, 64), 64)
    for p in m.parameters():
        p.requires_grad_(False)
    assert L.engine_refusal(m, FakeHip(64, 64), 64) is None
    assert L.engine_refusal(m, FakeHip(16, 16), 64, spatial_scale=4) is None            # a 1/4-resolution feature of a 64-multiple frame
    assert "multiples of 64" in L.engine_refusal(m, FakeHip(72, 64), 64)
    m.train()
    assert L.engine_refusal(m, FakeHip(64, 64), 64) == "module in training mode"
    with torch.no_grad():
        m.eval()
        for p in m.parameters():
            p.requires_grad_(True)
        assert L.engine_refusal(m, FakeHip(64, 64), 64) is None            # no gradients wanted: the engines serve it


def test_plane_graph_lets_the_first_reader_finalise_a_segment(monkeypatch):
    """PlaneGraph._fuse_single_reader_segments on descriptors only (CPU tensors, nothing is launched): a produced segment whose
    FIRST reader is a convolution over exactly that segment gets its gradient planes from that reader's transposed launch
    (`mask` + `out_planes`, and `add` = the float32 sum when other readers contributed before it); a concatenation read as a
    whole by its first reader, a segment that feeds predict_flow first, and a linear segment keep / lose what they should."""
    from understanding_flow_robustness_amd.plane_graph import PlaneGraph
    conv = lambda cin, cout, k=3: (torch.randn(cout, cin, k, k) * 0.1, torch.zeros(cout))
    monkeypatch.setenv("UFR_GRAPH_FUSE_FINALIZE", "1")
    g = PlaneGraph(1, "cpu")
    for name, s, chunks in (("in0", 0, 1), ("a", 0, 2), ("cat", 1, 5), ("b", 2, 4), ("ic", 1, 1)):
        g.buffer(name, 32 >> s, 64 >> s, chunks)
    g.input("in0", 6)
    g.conv(*conv(6, 64), ("in0", 0, 1), ("a", 0))                          # 0: a <- in0           (one reader: conv 1)
    g.conv(*conv(64, 64), ("a", 0, 2), ("cat", 0), stride=2)               # 1: cat[0:2] <- a      (readers: conv 2 exact, then conv 4 whole)
    g.conv(*conv(64, 128), ("cat", 0, 2), ("b", 0), stride=2)              # 2: b <- cat[0:2]      (first reader of b: predict_flow)
    g.predict_flow(torch.nn.Conv2d(128, 2, 3, 1, 1), ("b", 0, 4), "flow2")
    g.up_flow(torch.nn.ConvTranspose2d(2, 2, 4, 2, 1, bias=False), "flow2", ("cat", 4))
    dw = torch.randn(128, 64, 4, 4) * 0.1
    g.deconv(dw, torch.zeros(64), ("b", 0, 4), ("cat", 2))                 # 3: cat[2:4] <- b      (first reader: conv 4 over the WHOLE cat)
    g.conv(*conv(162, 32), ("cat", 0, 5), ("ic", 0), slope=1.0,            # 4: ic <- cat          (linear; read by predict_flow only)
           in_segments=[(0, 64, 0), (64, 64, 64), (128, 2, 128)])
    g.predict_flow(torch.nn.Conv2d(32, 2, 3, 1, 1), ("ic", 0, 1), "flow1")
    g.output("flow1")
    g.build()
    convs = [op for op in g.ops if op["kind"] == "conv"]
    assert [bool(op.get("finalized")) for op in convs] == [True, True, False, False, False]
    d1, d2 = convs[1]["bwd"].launch.desc, convs[2]["bwd"].launch.desc
    # conv 1's transposed launch is `a`'s only contribution: planes + mask, no sum
    assert d1.out_planes == convs[0]["gz"].t.data_ptr() and d1.mask == g.bufs["a"].planes.t.data_ptr() and not d1.add and not d1.out_f32
    # conv 2's is the LAST of two contributions to cat[0:2] (conv 4's came first in the backward): it adds the sum and finalises
    assert d2.out_planes == convs[1]["gz"].t.data_ptr() and d2.add == g.bufs["cat"].grad.t.data_ptr() and not d2.out_f32
    assert d2.mask == g.bufs["cat"].planes.t.data_ptr() and d2.mask_chunk0 == 0
    # the others still accumulate into their source's float32 sum
    for op in (convs[0], convs[3], convs[4]):
        d = op["bwd"].launch.desc
        assert d.out_f32 and not d.out_planes
    monkeypatch.setenv("UFR_GRAPH_FUSE_FINALIZE", "0")
    g0 = PlaneGraph(1, "cpu")
    g0.buffer("in0", 8, 8, 1); g0.buffer("a", 8, 8, 2); g0.buffer("b", 8, 8, 2)
    g0.input("in0", 6)
    g0.conv(*conv(6, 64), ("in0", 0, 1), ("a", 0))
    g0.conv(*conv(64, 64), ("a", 0, 2), ("b", 0))
    g0.build()
    assert not any(op.get("finalized") for op in g0.ops)


def test_igemm_host_arithmetic_runs_up_to_the_launch_without_a_device():
    """`ufr_igemm`'s host side -- descriptor checks, the phase / tap-run tables, the split-K slicing -- on REAL descriptors of every
    geometry the engines build (stride-1 and stride-2 convolutions, their data gradients with 1 + 2 + 2 + 4-tap phases and split-K,
    both deconvolution launches), on a box without a GPU: the call must get as far as asking for the device and fail THERE
    (UFR_ELAUNCH), never earlier and never by touching an operand.  This is the test `make -C csrc sanitize` (host ASan / UBSan
    build of the library) is pointed at: a descriptor field read past its struct, a tap table overrun or an overflowing extent
    computation aborts it."""
    import ctypes as C

    import pytest

    from understanding_flow_robustness_amd import _lib as L
    lib = L.lib()
    if lib.ufr_device_count() > 0:
        pytest.skip("a HIP device is present: these descriptors point at host memory and must not be launched")
    g = torch.Generator().manual_seed(3)
    B, H, W = 2, 12, 20
    x = ig.Planes(B, H, W, 4, "cpu")                                        # 128 channels
    w3 = torch.randn(96, 128, 3, 3, generator=g)
    w3s2 = torch.randn(128, 128, 3, 3, generator=g)
    wd = torch.randn(128, 64, 4, 4, generator=g)
    cases = []
    # Conv2d(128, 96, 3, 1, 1) forward, single pass and split-K with the reduce launch behind it
    wi = ig.conv_forward_weights(w3, 1, 1)
    out = ig.Planes(B, H, W, 3, "cpu")
    bias = torch.zeros(wi.Npad)
    for variant in (2, 4, 5, 6, 7, 8):
        cases.append(ig.make_launch(wi, x, 0, (H, W), (H, W), out_planes=out, bias=bias, variant=variant))
    ws = torch.empty(4 * B * H * W * wi.Npad)
    cases.append(ig.make_launch(wi, x, 0, (H, W), (H, W), out_planes=out, bias=bias, splitk=4, ws=ws, variant=6))
    # its data gradient (stride 1) with an addend and a mask
    wb = ig.conv_backward_weights(w3, 1, 1)
    gy, gx = ig.Planes(B, H, W, 3, "cpu"), ig.GradSum(B, H, W, 4, "cpu")
    cases.append(ig.make_launch(wb, gy, 0, (H, W), (H, W), out_f32=gx, add=gx, mask=x, variant=6))
    # Conv2d(128, 128, 3, 2, 1): forward on the half grid, data gradient as four phases of 1 + 2 + 2 + 4 taps, split 3 ways
    wf2, wb2 = ig.conv_forward_weights(w3s2, 2, 1), ig.conv_backward_weights(w3s2, 2, 1)
    half = ig.Planes(B, H // 2, W // 2, 4, "cpu")
    cases.append(ig.make_launch(wf2, x, 0, (H // 2, W // 2), (H // 2, W // 2), out_planes=half, bias=torch.zeros(wf2.Npad), variant=6))
    assert sorted(len(t) for _, _, t in wb2.phases) == [1, 2, 2, 4]
    ws2 = torch.empty(4 * 3 * B * (H // 2) * (W // 2) * wb2.Npad)
    cases.append(ig.make_launch(wb2, half, 0, (H // 2, W // 2), (H, W), out_f32=gx, splitk=3, ws=ws2, variant=6))
    # ConvTranspose2d(128, 64, 4, 2, 1): forward (four phases onto the fine grid) and its data gradient (16 taps, stride-2 reads)
    wdf, wdb = ig.deconv_forward_weights(wd, 1), ig.deconv_backward_weights(wd, 1)
    fine = ig.Planes(B, 2 * H, 2 * W, 2, "cpu")
    cases.append(ig.make_launch(wdf, x, 0, (H, W), (2 * H, 2 * W), out_planes=fine, bias=torch.zeros(wdf.Npad), variant=2))
    cases.append(ig.make_launch(wdb, fine, 0, (H, W), (H, W), out_f32=gx, variant=2))
    for launch in cases:
        rc = lib.ufr_igemm(C.byref(launch.desc), None)
        assert rc == -3, (rc, lib.ufr_last_error())
        assert b"no current device" in lib.ufr_last_error() or b"igemm" in lib.ufr_last_error()
    # and the refusals in front of it
    bad = cases[0].desc
    keep = bad.splitk
    bad.splitk = 65
    assert lib.ufr_igemm(C.byref(bad), None) == -1 and b"split" in lib.ufr_last_error()
    bad.splitk = keep


def test_frozen_module_deep_copy_keeps_the_callers_flags_and_workspace_caches_are_lru():
    """(a) `freeze_parameters` records the caller's `requires_grad` flags once per module; a deep copy of the frozen module
    carries the record re-keyed onto ITS parameters, so `restore_parameters(copy)` restores the copy (it used to get an empty record
    and stayed frozen for good); (b) the module-level workspace caches evict the least recently used entry instead of being cleared."""
    import copy

    from understanding_flow_robustness_amd import _lib as L
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.Conv2d(4, 2, 3))
    net[1].weight.requires_grad_(False)                                    # the caller froze one tensor himself
    want = [p.requires_grad for p in net.parameters()]
    L.freeze_parameters(net)
    assert not any(p.requires_grad for p in net.parameters())
    twin = copy.deepcopy(net)
    assert not any(p.requires_grad for p in twin.parameters())
    rec = twin.__dict__[L._GRAD_FLAGS_ATTR]
    assert {id(p) for p in twin.parameters()} == set(rec) and all(rec[id(p)][0] is p for p in twin.parameters())
    L.restore_parameters(twin)
    assert [p.requires_grad for p in twin.parameters()] == want
    assert not any(p.requires_grad for p in net.parameters())              # the original is still frozen ...
    L.restore_parameters(net)
    assert [p.requires_grad for p in net.parameters()] == want             # ... until its own restore
    c = L.LruDict(3)
    for k in "abc":
        c[k] = k.upper()
    assert c.get("a") == "A"                                               # a hit makes "a" the most recently used
    c["d"] = "D"
    assert sorted(c) == ["a", "c", "d"] and c.get("b") is None
    c["c"] = "C2"                                                          # an existing key never evicts
    assert len(c) == 3


def test_round6_host_switches(monkeypatch):
    """Host logic added in round 6, no device needed: RAFT's precision is an explicit opt-in (`args.mixed_precision` AND the environment
    -- models/utils_model.py:51 sets the flag for every non-"adv" RAFT, the build stays float32 unless asked twice); `igemm.products` nests
    and restores; the tuning key carries the row grid when the caller gives it and falls back to the short form; the split-plane lookup
    and the dense adjoint decline CPU tensors (the callers then take the other path or raise: there is no CPU fallback)."""
    import pytest
    from argparse import Namespace

    from understanding_flow_robustness_amd.flownets.raft import RAFT
    from understanding_flow_robustness_amd.flownets.raft_corr import AltCorrPlanes, alt_dense_adjoint_served
    net = RAFT(Namespace(small=False, mixed_precision=False))
    monkeypatch.delenv("UFR_RAFT_PRECISION", raising=False)
    assert net.products() == 6
    net.args.mixed_precision = True
    assert net.products() == 6                                             # the flag alone
    for mode, want in (("bf16", 1), ("bf16x3", 3), ("fp32", 6), ("FLOAT32", 6)):
        monkeypatch.setenv("UFR_RAFT_PRECISION", mode)
        assert net.products() == want
    net.args.mixed_precision = False
    monkeypatch.setenv("UFR_RAFT_PRECISION", "bf16")
    assert net.products() == 6                                             # the environment alone
    monkeypatch.setenv("UFR_RAFT_PRECISION", "fp16")
    with pytest.raises(ValueError):
        net.products()
    assert ig._PRODUCTS[-1] == 6
    with ig.products(1):
        assert ig._PRODUCTS[-1] == 1
        with ig.products(3):
            assert ig._PRODUCTS[-1] == 3
        assert ig._PRODUCTS[-1] == 1
    assert ig._PRODUCTS == [6]
    with pytest.raises(ValueError):
        ig.products(2)
    wi = ig.conv_forward_weights(torch.randn(64, 32, 3, 3), 1, 1)
    kw = dict(out_planes=object())
    short, long_ = ig.launch_signature(wi, 7680, kw), ig.launch_signature(wi, 7680, kw, (48, 160))
    assert long_ == short + "_g48x160" and short.startswith("M7680_N64_KC1_t9_s11_planes")
    monkeypatch.setattr(ig, "_TUNING", {short: (4, 2), long_: (8, 1)})
    assert ig.tuned(wi, 7680, kw, 6, 3, rows=(48, 160)) == (8, 1)          # the long form first
    assert ig.tuned(wi, 7680, kw, 6, 3, rows=(24, 320)) == (4, 2)          # another grid: the short form
    assert ig.tuned(wi, 7680, kw, 6, 3) == (4, 2)
    monkeypatch.setattr(ig, "_TUNING", {})
    assert ig.tuned(wi, 7680, kw, 6, 3, rows=(48, 160)) == (6, 3)          # nothing swept: the engine's own choice
    f1 = torch.zeros(1, 8, 16, 256)
    f2 = [torch.zeros(1, 8 >> l, 16 >> l, 256) for l in range(4)]
    assert not AltCorrPlanes.served(f1, f2, 4) and not alt_dense_adjoint_served(f1, f2, 4)

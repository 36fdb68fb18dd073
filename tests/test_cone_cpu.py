"""Cone-of-influence arithmetic (understanding_flow_robustness_amd/cone.py, mirrored by csrc/window.hip):
running a convolutional prefix on the window the arithmetic picks must reproduce the full-frame features
everywhere the patch can reach, and the full-frame gradient on every patch pixel -- at every placement,
including the frame's corners and edges.  Plain torch on the CPU; no HIP call."""
import itertools

import pytest
import torch
import torch.nn as nn

from understanding_flow_robustness_amd.cone import ConeSpec

SPEC = ConeSpec(layers=((7, 2, 3), (5, 2, 2), (5, 2, 2)), taps=(1, 2), frames=(1, 2))     # FlowNetC conv1-3


def _prefix(spec, width=4, seed=0):
    torch.manual_seed(seed)
    convs, cin = [], 3
    for k, s, p in spec.layers:
        convs.append(nn.Sequential(nn.Conv2d(cin, width, k, s, p), nn.LeakyReLU(0.1)))
        cin = width
    net = nn.ModuleList(convs).double()

    def encode(x):
        outs = []
        for i, c in enumerate(net):
            x = c(x)
            outs.append(x)
        return [outs[t] for t in spec.taps]
    return encode


def _rim(n_full, o, w, m):
    """Cells [a, b) of a window axis that are exact (window.hip: rim only at interior edges)."""
    return (m if o > 0 else 0), (w - m if o + w < n_full else w)


def test_margins_flownetc():
    assert SPEC.margins() == [2, 2, 2]
    assert SPEC.total_stride == 8 and SPEC.level_stride(1) == 4


# a second chain (3x3 convolutions, one stride-2 layer, taps after the first and the last layer) checks that nothing is
# specific to FlowNetC's kernel sizes
SPEC_B = ConeSpec(layers=((3, 1, 1), (3, 2, 1), (3, 1, 1)), taps=(0, 2), frames=(1, 2))


@pytest.mark.parametrize("spec", [SPEC, SPEC_B], ids=["flownetc", "3x3-chain"])
@pytest.mark.parametrize("ext", [(9, 9), (17, 5), (1, 1)])
def test_windowed_prefix_equals_full(ext, spec):
    H, W = 128, 160
    eh, ew = ext
    encode = _prefix(spec)
    wh, ww = spec.window_size(eh, H), spec.window_size(ew, W)
    assert wh < H and ww < W
    ys = sorted({0, 1, 7, 13, 22, H - eh - 9, H - eh - 1, H - eh})
    xs = sorted({0, 3, 8, 31, 40, W - ew - 8, W - ew - 2, W - ew})
    g = torch.Generator().manual_seed(1)
    for y, x in itertools.product(ys, xs):
        base = torch.rand(1, 3, H, W, generator=g, dtype=torch.float64)
        img = base.clone()
        img[:, :, y:y + eh, x:x + ew] = torch.rand(1, 3, eh, ew, generator=g, dtype=torch.float64)
        img.requires_grad_(True)
        cached = [f.detach().clone() for f in encode(base)]           # features of the previous iteration
        want = encode(img)
        assert spec.need(y, y + eh - 1, H)[1] * spec.total_stride <= wh and spec.need(x, x + ew - 1, W)[1] * spec.total_stride <= ww
        oy, ox = spec.origin(y, y + eh - 1, H, wh), spec.origin(x, x + ew - 1, W, ww)
        assert oy % spec.total_stride == 0 and ox % spec.total_stride == 0 and 0 <= oy <= H - wh and 0 <= ox <= W - ww
        xw = img.detach()[:, :, oy:oy + wh, ox:ox + ww].clone().requires_grad_(True)
        got_w = encode(xw)
        g_full = [torch.randn(f.shape, generator=g, dtype=torch.float64) for f in want]
        g_win = []
        for f_w, f_c, f_want, gf, t, m in zip(got_w, cached, want, g_full, spec.taps, spec.tap_margins()):
            ls = spec.level_stride(t)
            a0, b0 = _rim(H // ls, oy // ls, wh // ls, m)
            a1, b1 = _rim(W // ls, ox // ls, ww // ls, m)
            pasted = f_c.clone()
            pasted[:, :, oy // ls + a0:oy // ls + b0, ox // ls + a1:ox // ls + b1] = f_w.detach()[:, :, a0:b0, a1:b1]
            assert torch.allclose(pasted, f_want.detach(), rtol=0, atol=1e-12), (y, x, t)
            gw = torch.zeros_like(f_w)
            gw[:, :, a0:b0, a1:b1] = gf[:, :, oy // ls + a0:oy // ls + b0, ox // ls + a1:ox // ls + b1]
            g_win.append(gw)
        gi, = torch.autograd.grad(want, img, g_full)
        gxw, = torch.autograd.grad(got_w, xw, g_win)
        a = gi[:, :, y:y + eh, x:x + ew]
        b = gxw[:, :, y - oy:y - oy + eh, x - ox:x - ox + ew]
        assert torch.allclose(a, b, rtol=0, atol=1e-12), (y, x)


def test_window_size_covers_every_placement():
    for size, ext in ((384, 51), (1280, 51), (128, 20)):
        win = SPEC.window_size(ext, size)
        for lo in range(0, size - ext + 1):
            n_lo, cnt = SPEC.need(lo, lo + ext - 1, size)
            assert cnt * 8 <= win
            o = SPEC.origin(lo, lo + ext - 1, size, win)
            assert o <= n_lo * 8 and n_lo * 8 + cnt * 8 <= o + win
    assert SPEC.window_size(51, 1280) <= 144

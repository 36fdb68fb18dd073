"""Cone-of-influence bookkeeping for the windowed encoder of the patch attack (csrc/window.hip).

patch_attacks/main.py:537-600 keeps `mask * gradient` only and changes only masked pixels between the
iterations of one attack() call.  For a convolutional prefix of the network this bounds what has to be
recomputed: the prefix's outputs (and adjoints) matter only inside the patch's forward cone.  This
module holds the integer arithmetic -- the same as the device kernel `cone_axis` -- that sizes the
window; the origin itself is computed on the device so a captured graph follows new placements.
"""
from __future__ import annotations

from dataclasses import dataclass

from . import _lib as L


@dataclass(frozen=True)
class ConeSpec:
    """A convolutional prefix: `layers` = ((kernel, stride, pad), ...) input to output; `taps` = indices
    of the layers whose outputs leave the prefix (sorted); `frames[i]` = 2 when the rest of the network
    reads tap i for both frames, 1 for the first frame only.  The LAST tap must have frames = 2."""
    layers: tuple
    taps: tuple
    frames: tuple

    @property
    def total_stride(self):
        s = 1
        for _, st, _ in self.layers:
            s *= st
        return s

    def level_stride(self, layer):
        s = 1
        for _, st, _ in self.layers[:layer + 1]:
            s *= st
        return s

    def margins(self):
        """Per layer: cells next to an INTERIOR window edge whose zero-padded windowed value differs from
        the full-image one (left/top and right/bottom propagate differently; the larger is used)."""
        gl = gr = 0
        out = []
        for k, s, p in self.layers:
            gl = -(-(gl + p) // s)
            gr = max(-(-(gr + k - 1 - p - (s - 1)) // s), 0)
            out.append(max(gl, gr))
        return out

    def tap_margins(self):
        m = self.margins()
        return tuple(m[t] for t in self.taps)

    def to_c(self):
        ch = L.ConeChain()
        ch.n_layers = len(self.layers)
        for i, (k, s, p) in enumerate(self.layers):
            ch.kernel[i], ch.stride[i], ch.pad[i] = k, s, p
        ch.n_taps = len(self.taps)
        for i, (t, m) in enumerate(zip(self.taps, self.tap_margins())):
            ch.tap_layer[i], ch.tap_margin[i] = t, m
        return ch

    # -------------------------------------------------------------------------- host mirror of cone_axis
    def cone(self, lo, hi, size):
        """[lo, hi] (inclusive input pixels) -> list of per-layer (lo, hi, n_out) cones."""
        out, n = [], size
        for k, s, p in self.layers:
            n_out = (n + 2 * p - k) // s + 1
            lo = max(-(-(lo + p - (k - 1)) // s), 0)
            hi = min((hi + p) // s, n_out - 1)
            n = n_out
            out.append((lo, hi, n_out))
        return out

    def need(self, lo, hi, size):
        """Needed window on one axis: (first cell, cell count) in units of the deepest level."""
        total, cells = self.total_stride, size // self.total_stride
        if hi < lo:
            return 0, 0
        cones, margins = self.cone(lo, hi, size), self.tap_margins()
        n_lo, n_hi = None, None
        for t, m in zip(self.taps, margins):
            per = total // self.level_stride(t)
            a, b = (cones[t][0] - m) // per, (cones[t][1] + m) // per
            n_lo = a if n_lo is None else min(n_lo, a)
            n_hi = b if n_hi is None else max(n_hi, b)
        n_lo, n_hi = max(n_lo, 0), min(n_hi, cells - 1)
        return n_lo, n_hi - n_lo + 1

    def origin(self, lo, hi, size, win):
        """The device kernel's choice of window origin (pixels) for a window of `win` pixels."""
        total, cells = self.total_stride, size // self.total_stride
        n_lo, cnt = self.need(lo, hi, size)
        if cnt == 0:
            return 0
        wcells = win // total
        o = n_lo - max(wcells - cnt, 0) // 2
        return min(max(o, 0), max(cells - wcells, 0)) * total

    def window_size(self, extent, size, slack=1):
        """Static window size (pixels) that fits a patch bounding box of `extent` pixels at ANY
        position on an axis of `size` pixels (+ `slack` cells), capped at the axis."""
        extent = min(extent, size)
        worst = max(self.need(lo, lo + extent - 1, size)[1] for lo in range(0, size - extent + 1))
        return min((worst + slack) * self.total_stride, size)

"""Host side of csrc/igemm.hip: plane buffers, pre-split weight images and the geometry descriptors that turn every
Conv2d / ConvTranspose2d block of models/FlowNetC.py:22-50 (models/submodules.py:18-46, :75-82) -- forward and data
gradient -- into one launch of `ufr_igemm` (float32-accurate on the bf16 matrix cores: three bf16 planes per operand,
six products).  No torch arithmetic happens here at run time: weights are packed once (they are frozen
during an attack), everything else is descriptors over device pointers.
"""
from __future__ import annotations

import ctypes as C

import os

import torch

from . import _lib as L

LEAKY = 0.1                  # models/submodules.py:33, :45, :81


def pad32(c: int) -> int:
    return (c + 31) // 32 * 32


def pad128(c: int) -> int:
    return (c + 127) // 128 * 128


def pad_n(c: int) -> int:
    """Output channels are padded to whole tiles: 128 columns, or 64 where that saves a quarter of the launch or more."""
    p64, p128 = (c + 63) // 64 * 64, pad128(c)
    return p64 if p64 * 4 <= p128 * 3 else p128


class Planes:
    """An activation in the engine's layout: bf16 [3][chunks][M][32], M = B*H*W pixels in (b, y, x) order, 32 channels
    per chunk, value = p0 + p1 + p2 exactly.  A torch.cat of the reference is a chunk offset into one wider buffer."""

    def __init__(self, B, H, W, chunks, device):
        self.B, self.H, self.W, self.chunks = int(B), int(H), int(W), int(chunks)
        self.M = self.B * self.H * self.W
        self.t = torch.zeros(3, self.chunks, self.M, 32, dtype=torch.bfloat16, device=device)
        self.plane_stride = self.chunks * self.M * 32       # (padding the planes apart -- 4 KB and 1 MB tried -- changes nothing:
                                                            #  5.460 / 5.465 / 5.462 ms in one call, gpurun r4_call36)

    def load_nchw(self, x: torch.Tensor, chunk0: int = 0, scale: float = 1.0, slope: float = 1.0, bias: torch.Tensor | None = None):
        """planes[chunk0 + c/32] = split(leaky(scale * x + bias[c])); x [B,C,H,W] float32 contiguous."""
        L.require_hip(x, "x")
        B, Cn, H, W = x.shape
        if (B, H, W) != (self.B, self.H, self.W) or chunk0 + pad32(Cn) // 32 > self.chunks or x.dtype != torch.float32:
            raise RuntimeError("Planes.load_nchw: shape mismatch")
        L.check(L.lib().ufr_nchw_to_planes(L.ptr(x), L.ptr(self.t), self.plane_stride, int(chunk0), B, Cn, H, W,
                                           float(scale), float(slope), L.ptr(bias) if bias is not None else None, L.stream()),
                "nchw -> planes")
        return self

    def load_rowmajor(self, src: torch.Tensor, chunk0: int = 0, scale: float = 1.0):
        """planes[chunk0 + k/32][m][k%32] = split(scale * src[m][k]) for a row-major float32 matrix src [M, K] (the operand of a
        GEMM reduced over the matrix's columns; rows = this buffer's pixels)."""
        L.require_hip(src, "src", contiguous=False)
        if src.dim() != 2 or src.stride(1) != 1 or src.dtype != torch.float32 or src.shape[0] != self.M or chunk0 + pad32(src.shape[1]) // 32 > self.chunks:
            raise RuntimeError("Planes.load_rowmajor: shape mismatch")
        L.check(L.lib().ufr_rowmajor_to_planes(L.ptr(src), src.stride(0), src.shape[0], src.shape[1], float(scale), L.ptr(self.t),
                                               self.plane_stride, int(chunk0), self.M, L.stream()), "row-major -> planes")
        return self

    def to_nchw(self, Cn: int, chunk0: int = 0, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.B, Cn, self.H, self.W, dtype=torch.float32, device=self.t.device)
        L.check(L.lib().ufr_chunks_to_nchw(L.ptr(self.t), self.plane_stride, None, int(chunk0), None, 0, L.ptr(out), self.B, Cn,
                                           self.H, self.W, 1.0, 1.0, L.stream()), "planes -> nchw")
        return out


class GradSum:
    """A gradient accumulator: float32 [chunks][M][32] in the same pixel / channel order as `Planes`."""

    def __init__(self, B, H, W, chunks, device):
        self.B, self.H, self.W, self.chunks = int(B), int(H), int(W), int(chunks)
        self.M = self.B * self.H * self.W
        self.t = torch.zeros(self.chunks, self.M, 32, dtype=torch.float32, device=device)

    def to_nchw(self, Cn: int, chunk0: int = 0, mask: Planes | None = None, mask_chunk0: int = 0, scale: float = 1.0,
                slope: float = LEAKY, out: torch.Tensor | None = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.B, Cn, self.H, self.W, dtype=torch.float32, device=self.t.device)
        L.check(L.lib().ufr_chunks_to_nchw(None, 0, L.ptr(self.t), int(chunk0), L.ptr(mask.t) if mask is not None else None,
                                           int(mask_chunk0), L.ptr(out), self.B, Cn, self.H, self.W, float(scale), float(slope),
                                           L.stream()), "gradient sum -> nchw")
        return out


# ---------------------------------------------------------------------------------------------------- weights
class WeightImage:
    """Pre-split weights of ONE launch: bf16 [3][total] with, per phase, a chunk-major [taps*KC][Npad][32] image at
    `offsets[z]`; `phases[z]` = (oy0, ox0, [(dy, dx)])."""

    def __init__(self, planes, offsets, phases, N, Npad, KC, geometry, C=None):
        self.planes, self.offsets, self.phases = planes, offsets, phases
        self.N, self.Npad, self.KC = N, Npad, KC
        self.C = C if C is not None else KC * 32          # real input channels (KC*32 includes the chunk padding)
        self.geometry = geometry        # dict(in_s, out_s): strides of the row grid -> input / output pixels
        self.k_order = K_ORDER          # how _pack ordered the K tiles

    def flops(self, rows: int) -> float:
        """Algorithmic FLOPs of one launch over `rows` tile rows: 2 * rows * taps * Cin * Cout, real channel counts."""
        return 2.0 * rows * sum(len(t) for _, _, t in self.phases) * self.C * self.N


def _split3(w: torch.Tensor) -> torch.Tensor:
    """float32 -> [3, n] bf16 planes with w == p0 + p1 + p2 exactly (round to nearest even at every step)."""
    w = w.reshape(-1).float()
    p0 = w.to(torch.bfloat16)
    r1 = w - p0.float()
    p1 = r1.to(torch.bfloat16)
    p2 = (r1 - p1.float()).to(torch.bfloat16)
    return torch.stack((p0, p1, p2)).contiguous()


# Order of a phase's K tiles (ufr_igemm_desc.k_order).  1: all taps of one 32-channel chunk, then the next chunk -- a tile's
# taps read overlapping pixels, so eight of nine reads of a 3x3 layer hit in L2 instead of streaming the activation again
# per tap (0 = tap-major, round 2's first form: 18.5 instead of 10.1 GB per iteration); the image and the descriptor agree.
K_ORDER = 1


def _pack(mats, device):
    """mats[z] = float32 [N, taps, C] (output channel, tap, input channel) per phase -> WeightImage arrays."""
    images, offsets, total = [], [], 0
    N, _, Cn = mats[0].shape
    npad, cpad = pad_n(N), pad32(Cn)
    for m in mats:
        taps = m.shape[1]
        w = torch.zeros(npad, taps, cpad, dtype=torch.float32, device=device)
        w[:N, :, :Cn] = m
        if K_ORDER:                      # [KC][taps][Npad][32]: the taps of a channel chunk are consecutive K tiles
            img = w.view(npad, taps, cpad // 32, 32).permute(2, 1, 0, 3).contiguous().view(-1)
        else:                            # [taps][KC][Npad][32]
            img = w.view(npad, taps * cpad // 32, 32).permute(1, 0, 2).contiguous().view(-1)
        offsets.append(total)
        total += img.numel()
        images.append(img)
    return _split3(torch.cat(images)), offsets, N, npad, cpad // 32, Cn


def _deconv_phases(kernel: int, padding: int):
    """out[2q + o] += w[k] * x[q + d]: the four phases of a stride-2 transposed convolution whose output is exactly twice
    its input (output_padding = 2 + 2*padding - kernel in {0, 1}); -> [(oy0, ox0, [(ky, kx, dy, dx)])]."""
    if not 0 <= 2 + 2 * padding - kernel <= 1:
        raise ValueError("stride-2 transposed convolution: kernel / padding do not double the size")
    phases = []
    for oy0 in (0, 1):
        for ox0 in (0, 1):
            ry, rx = (oy0 + padding) % 2, (ox0 + padding) % 2
            cy, cx = (oy0 + padding - ry) // 2, (ox0 + padding - rx) // 2
            phases.append((oy0, ox0, [(ky, kx, cy - (ky - ry) // 2, cx - (kx - rx) // 2)
                                      for ky in range(ry, kernel, 2) for kx in range(rx, kernel, 2)]))
    return phases


def _pair(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def conv_forward_weights(weight: torch.Tensor, stride: int, padding, dilation: int = 1) -> WeightImage:
    """Conv2d weight [N,C,kh,kw] -> the forward launch (rows = output grid); a dilation is just other tap offsets, a
    rectangular kernel (RAFT's 1x5 / 5x1 GRU convolutions, padding = (ph, pw)) just another tap list."""
    N, Cn, kh, kw = weight.shape
    ph, pw = _pair(padding)
    w = weight.detach().float()
    mat = w.permute(0, 2, 3, 1).reshape(N, kh * kw, Cn)
    taps = [(ky * dilation - ph, kx * dilation - pw) for ky in range(kh) for kx in range(kw)]
    if stride == 2 and dilation == 1 and kw >= 3:
        # a stride-2 row reads every other input pixel: the taps of one kernel row whose dx have the same parity read the SAME
        # pixel sequence shifted by whole cells, so they sit next to each other in K (dx = -2, 0, 2 | -1, 1 for a 5 x 5) and the
        # tap-reuse form (csrc/igemm.hip variant 7) stages each parity's pixels once per kernel row
        order = [ky * kw + kx for ky in range(kh) for par in (0, 1) for kx in range(kw) if (kx - pw) % 2 == par]
        mat, taps = mat[:, order], [taps[i] for i in order]
    planes, offsets, N, npad, KC, cr = _pack([mat], weight.device)
    return WeightImage(planes, offsets, [(0, 0, taps)], N, npad, KC, dict(in_s=stride, out_s=1), cr)


def conv1_packed_weights(weight: torch.Tensor) -> WeightImage:
    """Conv2d(3, N, 7, 2, 3) weight [N,3,7,7] -> the 8-tap stride-1 launch over the PACKED planes of `ufr_conv1_pack_planes`
    (csrc/igemm.hip): tap (a, b2) reads packed pixel (Y + a, X + 2 b2), a in 0..3, b2 in 0..1; its chunk holds channel
    j*12 + (c*2 + p)*2 + q = frame[c, 2 (Y + a - 2) + p, 2 (X + 2 b2 - 2 + j) + q], i.e. kernel tap (2a - 1 + p, 4 b2 - 1 + 2j + q)."""
    N, Cn, k, _ = weight.shape
    if (Cn, k) != (3, 7):
        raise ValueError("conv1_packed_weights: Conv2d(3, N, 7, 2, 3) only")
    w = weight.detach().float()
    mat = torch.zeros(N, 8, 32, dtype=torch.float32, device=weight.device)
    for a in range(4):
        for b2 in range(2):
            for j in range(2):
                for c in range(3):
                    for p in range(2):
                        for q in range(2):
                            ky, kx = 2 * a - 1 + p, 4 * b2 - 1 + 2 * j + q
                            if 0 <= ky < 7 and 0 <= kx < 7:
                                mat[:, a * 2 + b2, j * 12 + (c * 2 + p) * 2 + q] = w[:, c, ky, kx]
    taps = [(a, 2 * b2) for a in range(4) for b2 in range(2)]
    planes, offsets, n, npad, KC, _ = _pack([mat], weight.device)
    return WeightImage(planes, offsets, [(0, 0, taps)], n, npad, KC, dict(in_s=1, out_s=1), C=147.0 / 8.0)   # flops(): 147 real taps x channels


def planes_as_weights(p: Planes, N: int | None = None) -> WeightImage:
    """An activation's planes [3][chunks][M][32] ARE a weight image [3][KC][Npad = M][32] of a one-tap launch whose output
    channel n is pixel n: <x[p], y[q]> over the channels for all pixel pairs = RAFT's all-pairs correlation
    (models/raft/corr.py:57-64) as ONE igemm launch per frame pair.  M must be a multiple of 64."""
    if p.M % 64:
        raise ValueError("planes_as_weights: the pixel count must be a multiple of 64")
    planes = p.t.view(3, -1)
    wi = WeightImage(planes, [0], [(0, 0, [(0, 0)])], p.M if N is None else N, p.M, p.chunks, dict(in_s=1, out_s=1), C=p.chunks * 32.0)
    wi._keep = p
    return wi


def conv1_direct_weights(weight: torch.Tensor) -> torch.Tensor:
    """Conv2d(3, 64, 7, 2, 3) weight [64,3,7,7] -> the image `ufr_conv1_direct` (csrc/conv1_direct.hip) keeps in registers:
    bf16 [3 planes][7 ky][64 n][32 k], k = (kx >> 1) * 8 + (kx & 1) * 4 + c -- a K group of 8 = two adjacent input pixels x
    (3 channels + 1 zero); kx = 7 and c = 3 are zero."""
    N, Cn, k, _ = weight.shape
    if (N, Cn, k) != (64, 3, 7):
        raise ValueError("conv1_direct_weights: Conv2d(3, 64, 7, 2, 3) only")
    w = weight.detach().float()
    img = torch.zeros(7, 64, 32, dtype=torch.float32, device=weight.device)
    for kx in range(7):
        for c in range(3):
            img[:, :, (kx >> 1) * 8 + (kx & 1) * 4 + c] = w[:, c, :, kx].t()
    return _split3(img).view(3, 7, 64, 32).contiguous()


def conv1_packed_backward_weights(weight: torch.Tensor) -> WeightImage:
    """Data gradient of `conv1_packed_weights`' launch with respect to the PACKED planes: rows = the packed grid
    (H/2 + 3, W/2 + 2), gP(yp, xp)[n] = sum_t W_t[o, n] gy(yp - a_t, xp - 2 b2_t)[o]; N = 24 packed channels."""
    N, Cn, k, _ = weight.shape
    if (Cn, k) != (3, 7):
        raise ValueError("conv1_packed_backward_weights: Conv2d(3, N, 7, 2, 3) only")
    w = weight.detach().float()
    mat = torch.zeros(24, 8, N, dtype=torch.float32, device=weight.device)
    for a in range(4):
        for b2 in range(2):
            for j in range(2):
                for c in range(3):
                    for p in range(2):
                        for q in range(2):
                            ky, kx = 2 * a - 1 + p, 4 * b2 - 1 + 2 * j + q
                            if 0 <= ky < 7 and 0 <= kx < 7:
                                mat[j * 12 + (c * 2 + p) * 2 + q, a * 2 + b2, :] = w[:, c, ky, kx]
    taps = [(-a, -2 * b2) for a in range(4) for b2 in range(2)]
    planes, offsets, n, npad, KC, _ = _pack([mat], weight.device)
    return WeightImage(planes, offsets, [(0, 0, taps)], n, npad, KC, dict(in_s=1, out_s=1), C=147.0 * N / (8.0 * 24.0))


def conv_backward_weights(weight: torch.Tensor, stride: int, padding, dilation: int = 1) -> WeightImage:
    """Data gradient of Conv2d(weight [N,C,kh,kw], stride, padding, dilation): gx[C] from gy[N] (rows = the gy grid)."""
    N, Cn, kh, kw = weight.shape
    ph, pw = _pair(padding)
    w = weight.detach().float()
    if stride == 1:
        # gx[y] = sum_ky gy[y + p - d ky] * w[ky]: a convolution of gy with the transposed weights
        mat = w.permute(1, 2, 3, 0).reshape(Cn, kh * kw, N)
        taps = [(ph - ky * dilation, pw - kx * dilation) for ky in range(kh) for kx in range(kw)]
        planes, offsets, n, npad, KC, cr = _pack([mat], weight.device)
        return WeightImage(planes, offsets, [(0, 0, taps)], n, npad, KC, dict(in_s=1, out_s=1), cr)
    k, padding = kh, ph
    if kh != kw or ph != pw:
        raise NotImplementedError("data gradient of a strided convolution: square kernels only")
    if stride != 2 or dilation != 1:
        raise NotImplementedError("data gradient: stride 1 (any dilation) or 2")
    phases = _deconv_phases(k, padding)          # gx = conv_transpose(gy, weight): 'input' channels N, output channels C
    mats = [torch.stack([w[:, :, ky, kx].t() for ky, kx, _, _ in taps], dim=1) for _, _, taps in phases]   # [C, taps, N]
    planes, offsets, n, npad, KC, cr = _pack(mats, weight.device)
    return WeightImage(planes, offsets, [(oy, ox, [(dy, dx) for _, _, dy, dx in taps]) for oy, ox, taps in phases], n, npad, KC,
                       dict(in_s=1, out_s=2), cr)


def deconv_forward_weights(weight: torch.Tensor, padding: int) -> WeightImage:
    """ConvTranspose2d(Cin, Cout, k, 2, padding) weight [Cin,Cout,k,k] -> the forward launch (rows = the coarse grid)."""
    cin, cout, k, _ = weight.shape
    w = weight.detach().float()
    phases = _deconv_phases(k, padding)
    mats = [torch.stack([w[:, :, ky, kx].t() for ky, kx, _, _ in taps], dim=1) for _, _, taps in phases]   # [Cout, taps, Cin]
    planes, offsets, n, npad, KC, cr = _pack(mats, weight.device)
    return WeightImage(planes, offsets, [(oy, ox, [(dy, dx) for _, _, dy, dx in taps]) for oy, ox, taps in phases], n, npad, KC,
                       dict(in_s=1, out_s=2), cr)


def deconv_backward_weights(weight: torch.Tensor, padding: int) -> WeightImage:
    """Data gradient of that ConvTranspose2d: gx[Cin] on the coarse grid = Conv2d(gy, weight as [Cin,Cout,k,k], stride 2,
    padding) (rows = the coarse grid, taps read the fine gradient)."""
    cin, cout, k, _ = weight.shape
    w = weight.detach().float()
    mat = w.permute(0, 2, 3, 1).reshape(cin, k * k, cout)
    taps = [(ky - padding, kx - padding) for ky in range(k) for kx in range(k)]
    planes, offsets, n, npad, KC, cr = _pack([mat], weight.device)
    return WeightImage(planes, offsets, [(0, 0, taps)], n, npad, KC, dict(in_s=2, out_s=1), cr)


# ---------------------------------------------------------------------------------------------------- launches
class Launch:
    """One prepared `ufr_igemm` call: the descriptor plus references that keep its tensors alive."""

    def __init__(self, desc, keep):
        self.desc, self._keep = desc, keep

    def __call__(self):
        L.check(L.lib().ufr_igemm(C.byref(self.desc), L.stream()), "igemm")

    def algorithmic_bytes(self) -> float:
        """What one launch has to move if every operand crosses HBM exactly once (the yardstick its PMC traffic is read
        against): the input pixels under the row grid (three bf16 planes of KC chunks), the weight image, the output (three
        planes or one fp32 value per element), the fp32 addend and the mask plane; split-K slabs count as waste, not work."""
        d = self.desc
        rows = d.B * d.Hr * d.Wr
        pix_in = min(d.B * d.Hi * d.Wi, rows * d.in_sy * d.in_sx)
        taps = sum(d.phase[z].ntaps for z in range(d.nphase))
        out_el = rows * d.nphase * ((d.tail_n0 if d.tail else d.N) + 31) // 32 * 32
        b = pix_in * d.KC * 32 * 6 + taps * d.KC * d.Npad * 32 * 6
        b += out_el * ((4 if d.add else 0) + (2 if d.mask else 0))
        nch = out_el // (rows * d.nphase * 32)                  # output chunks; planes / fp32 may each cover a part of them
        pl = min(nch, d.planes_chunks) if d.planes_chunks else nch
        b += rows * d.nphase * 32 * ((6 * pl if d.out_planes else 0) + (4 * (nch - min(nch, d.f32_first_chunk)) if d.out_f32 else 0))
        if d.tail:
            b += rows * (d.N - d.tail_n0) * 4
        return float(b)


# ---- per-launch tuning (round 5) ------------------------------------------------------------------------------------------
# Every engine picks a launch's kernel form and split-K factor by a rule of thumb (`splitk_for` + its own default variant); on the
# small grids of the one-pair configs (RAFT's 48 x 160 update block, FlowNet2's sub-networks at 448 x 1024) the best pair
# depends on the launch.  `tools/sweep_engine_launches.py` times every distinct launch of a config with every (variant, split-K)
# pair on the step's own buffers and writes the winners to `igemm_tuning.json`, keyed by `launch_signature`; `tuned()` is asked
# by every engine right after its rule of thumb.  UFR_IGEMM_TUNING=0 switches the table off (the sweep tool does, to see the rule).
_TUNING = None
# Split-K without the second launch (csrc/igemm.hip `tickets`, round 6): the last workgroup to arrive at a tile adds the tile's slabs in
# ascending order and runs the epilogue -- bit-identical to the reduce kernel (tests/test_igemm_gpu.py).  BUILT AND MEASURED SLOWER on every
# config, both forms in one call (profiles/r6_fused_splitk_ab.jsonl: RAFT 15.29 -> 18.14 ms, FlowNet2 9.45 -> 11.46, PWC-Net 16.20 -> 17.30,
# FlowNetC 4.29 -> 4.66): every slice's workgroup pays an agent-scope release (a write-back of its XCD's L2) where the two-launch form pays
# one kernel boundary.  It stays an opt-in: UFR_IGEMM_FUSE_REDUCE=1, or `fuse_reduce=True` per launch.
FUSE_REDUCE = os.environ.get("UFR_IGEMM_FUSE_REDUCE", "0") == "1"
# Products per float32 product of the launches being built: 6 everywhere; RAFT's engines build theirs inside `with products(1):` when
# the caller opted into reduced precision (flownets/raft.py `raft_precision`).
_PRODUCTS = [6]


class products:
    """`with igemm.products(n):` -- launches made inside compute n (6, 3 or 1) bf16 products per float32 product."""
    def __init__(self, n: int):
        if n not in (6, 3, 1):
            raise ValueError("igemm.products: 6, 3 or 1")
        self.n = n

    def __enter__(self):
        _PRODUCTS.append(self.n)

    def __exit__(self, *exc):
        _PRODUCTS.pop()


RECORD = None            # a list while tools/sweep_engine_launches.py builds a step: every make_launch() appends its arguments


def launch_signature(wi: "WeightImage", M: int, kw: dict, rows=None) -> str:
    """What decides a launch's best form: rows, padded columns, K chunks, taps per phase, strides, and how the result leaves; with
    `rows` = (Hr, Wr) also the row grid's shape (round 6, ADVICE r5: variant 8 needs stride-1 3 x 3 launches, variant 7 a grid of at
    least 22 columns -- two launches with equal M and different grids are different launches; new sweeps write the long form)."""
    out = ("slabs" if kw.get("no_reduce") else "rowmajor" if kw.get("out_rowmajor") is not None else
           ("planes" if kw.get("out_planes") is not None else "") + ("f32" if kw.get("out_f32") is not None else ""))
    out += ("+add" if kw.get("add") is not None else "") + ("+mask" if kw.get("mask") is not None else "") + ("+tail" if kw.get("tail") is not None else "")
    taps = "-".join(str(len(t)) for _, _, t in wi.phases)
    grid = f"_g{int(rows[0])}x{int(rows[1])}" if rows is not None else ""
    return f"M{M}_N{wi.Npad}_KC{wi.KC}_t{taps}_s{wi.geometry['in_s']}{wi.geometry['out_s']}_{out}{grid}"


def _tuning_table() -> dict:
    global _TUNING
    if _TUNING is None:
        _TUNING = {}
        # (UFR_IGEMM_TUNING_FILE: another table for a same-call A/B of two candidate tables)
        path = os.environ.get("UFR_IGEMM_TUNING_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "igemm_tuning.json")
        if os.environ.get("UFR_IGEMM_TUNING", "1") != "0" and os.path.exists(path):
            import json
            with open(path) as f:
                raw = json.load(f)
            if raw.get("_arch", "gfx950") == "gfx950":          # the table is this chip's (the library is built for gfx950 only)
                _TUNING = {k: tuple(v) for k, v in raw.items() if not k.startswith("_")}
    return _TUNING


def variant_fallbacks() -> int:
    """Launches since load that asked for the direct 3 x 3 form (variant 8) or the tap-reuse form (variant 7) and ran as a plain tile form
    because their geometry is not covered (csrc/igemm.hip counts them): a tuning-table entry that reached the wrong launch shows here."""
    return int(L.lib().ufr_igemm_variant_fallbacks())


def tuned(wi: "WeightImage", M: int, kw: dict, variant: int, splitk: int, rows=None):
    """(variant, splitk) of the tuning table for this launch, else the engine's own choice.  A launch whose twin with (without) the
    epilogue's `add` was swept takes the twin's entry: the addend is one more 32-byte read per output row of the epilogue, nothing
    the choice of form depends on (PlaneGraph drops the `add` of a segment's first writer: plane_graph._plan_first_writers)."""
    table, sig = _tuning_table(), launch_signature(wi, M, kw)
    if rows is not None and launch_signature(wi, M, kw, rows) in table:       # the long form (with the row grid) first
        return table[launch_signature(wi, M, kw, rows)]
    if sig in table:
        return table[sig]
    head, _, out = sig.rpartition("_")
    twin = out.replace("+add", "") if "+add" in out else (out.replace("+mask", "+add+mask") if "+mask" in out else
                                                         (out.replace("+tail", "+add+tail") if "+tail" in out else out + "+add"))
    return table.get(f"{head}_{twin}", (variant, splitk))


def make_launch(wi: WeightImage, x: Planes, in_chunk0: int, rows_hw, out_hw, *, out_planes: Planes | None = None,
                out_chunk0: int = 0, out_f32: GradSum | None = None, out_f32_chunk0: int = 0, bias: torch.Tensor | None = None,
                add: GradSum | None = None, add_chunk0: int = 0, mask: Planes | None = None, mask_chunk0: int = 0,
                slope: float = LEAKY, splitk: int = 1, ws: torch.Tensor | None = None, row_band=None, in_band=None,
                products: int | None = None, variant: int = 0, tail: GradSum | None = None, tail_n0: int = 0, tail_chunk0: int = 0,
                tail_accumulate: bool = False, out_rowmajor=None, planes_chunks: int = 0, f32_first_chunk: int = 0,
                no_reduce: bool = False, fuse_reduce: bool | None = None) -> Launch:
    """Descriptor for `wi` applied to the chunks [in_chunk0, in_chunk0 + KC) of `x`.
    rows_hw = (Hr, Wr) row grid; out_hw = (Ho, Wo) output grid.  row_band = (origins int32 tensor, element stride, divisor):
    tail / tail_n0: the launch's output columns >= tail_n0 are a LATER layer's partial sum over these input chunks and leave raw
    into `tail` (chunk `tail_chunk0` on; `tail_accumulate`: added onto it); that layer's own launch passes it as `add`.
    the row grid's columns start at origins[b*stride] // divisor.  in_band = (origins, stride, divisor, width): input
    columns outside [origin, origin + width) read as zero.  bias given -> forward epilogue (bias + LeakyReLU)."""
    if RECORD is not None:
        RECORD.append(dict(wi=wi, x=x, in_chunk0=in_chunk0, rows=tuple(rows_hw), out_hw=tuple(out_hw), splitk=splitk, variant=variant,
                           kw=dict(out_planes=out_planes, out_chunk0=out_chunk0, out_f32=out_f32, out_f32_chunk0=out_f32_chunk0, bias=bias,
                                   add=add, add_chunk0=add_chunk0, mask=mask, mask_chunk0=mask_chunk0, slope=slope, row_band=row_band,
                                   in_band=in_band, tail=tail, tail_n0=tail_n0, tail_chunk0=tail_chunk0, tail_accumulate=tail_accumulate,
                                   out_rowmajor=out_rowmajor, planes_chunks=planes_chunks, f32_first_chunk=f32_first_chunk,
                                   no_reduce=no_reduce)))
    d = L.IgemmDesc()
    d.x, d.x_plane_stride, d.in_chunk0, d.KC = x.t.data_ptr(), x.plane_stride, int(in_chunk0), wi.KC
    if in_chunk0 + wi.KC > x.chunks:
        raise RuntimeError("igemm: the reduced chunks leave the input buffer")
    d.B, d.Hi, d.Wi = x.B, x.H, x.W
    d.in_sy = d.in_sx = int(wi.geometry["in_s"])
    if in_band is not None:
        d.in_x0, d.in_x0_stride, d.in_x0_div, d.in_xw = in_band[0].data_ptr(), int(in_band[1]), int(in_band[2]), int(in_band[3])
    d.w, d.w_plane_stride, d.Npad, d.N = wi.planes.data_ptr(), wi.planes.stride(0), wi.Npad, wi.N
    d.Hr, d.Wr = int(rows_hw[0]), int(rows_hw[1])
    if row_band is not None:
        d.row_x0, d.row_x0_stride, d.row_x0_div = row_band[0].data_ptr(), int(row_band[1]), int(row_band[2])
    d.Ho, d.Wo = int(out_hw[0]), int(out_hw[1])
    d.out_sy = d.out_sx = int(wi.geometry["out_s"])
    d.nphase = len(wi.phases)
    for z, ((oy0, ox0, taps), off) in enumerate(zip(wi.phases, wi.offsets)):
        ph = d.phase[z]
        if len(taps) > L.UFR_IGEMM_MAX_TAPS:
            raise RuntimeError("igemm: too many taps")
        ph.ntaps, ph.oy0, ph.ox0, ph.w_off = len(taps), oy0, ox0, off
        for t, (dy, dx) in enumerate(taps):
            ph.dy[t], ph.dx[t] = dy, dx
    Mout = x.B * d.Ho * d.Wo
    nch = pad32(wi.N) // 32
    if tail is not None:
        if tail_n0 % 32 or not 0 < tail_n0 < wi.N or tail.M != Mout or tail_chunk0 + (wi.N - tail_n0 + 31) // 32 > tail.chunks:
            raise RuntimeError("igemm: tail geometry")
        d.tail, d.tail_n0 = tail.t.data_ptr() + int(tail_chunk0) * tail.M * 32 * 4, int(tail_n0)
        d.tail_accumulate = 1 if tail_accumulate else 0
        nch = tail_n0 // 32                                       # chunks that reach the planes / fp32 output
    keep = [wi, x, bias, add, mask, out_planes, out_f32, ws, row_band, in_band, tail]
    if bias is not None:
        d.act, d.bias = 1, bias.data_ptr()
    d.slope = float(slope)
    if add is not None:
        if add.M != Mout or add_chunk0 + nch > add.chunks:
            raise RuntimeError("igemm: addend geometry")
        d.add, d.add_chunk0 = add.t.data_ptr(), int(add_chunk0)
    if mask is not None:
        if mask.M != Mout or mask_chunk0 + nch > mask.chunks:
            raise RuntimeError("igemm: mask geometry")
        d.mask, d.mask_chunk0 = mask.t.data_ptr(), int(mask_chunk0)
    if out_planes is not None:
        if out_planes.M != Mout or out_chunk0 + nch > out_planes.chunks:
            raise RuntimeError("igemm: output plane geometry")
        d.out_planes, d.out_plane_stride, d.out_chunk0 = out_planes.t.data_ptr(), out_planes.plane_stride, int(out_chunk0)
    if out_f32 is not None:
        if out_f32.M != Mout or out_f32_chunk0 + nch > out_f32.chunks:
            raise RuntimeError("igemm: fp32 output geometry")
        d.out_f32, d.out_f32_chunk0 = out_f32.t.data_ptr(), int(out_f32_chunk0)
    if out_rowmajor is not None:               # (tensor, element offset, leading dimension): fp32 [Mout][ld], element (pixel, n)
        rm, rm_off, ld = out_rowmajor
        if rm.dtype != torch.float32 or wi.N % 8 or ld < wi.N or ld % 4 or rm_off % 4 or rm.numel() < rm_off + (Mout - 1) * ld + wi.N:
            raise RuntimeError("igemm: row-major output geometry")
        d.out_rowmajor, d.out_ld = rm.data_ptr() + 4 * int(rm_off), int(ld)
        keep.append(rm)
    if planes_chunks or f32_first_chunk:       # with both outputs: planes for the chunks < planes_chunks, fp32 from f32_first_chunk on
        if out_planes is None or out_f32 is None:
            raise RuntimeError("igemm: planes_chunks / f32_first_chunk need both outputs")
        d.planes_chunks, d.f32_first_chunk = int(planes_chunks), int(f32_first_chunk)
    d.splitk = int(splitk)
    if splitk > 1:
        need = d.nphase * splitk * x.B * d.Hr * d.Wr * wi.Npad
        if ws is None or ws.numel() < need or ws.dtype != torch.float32:
            raise RuntimeError(f"igemm: split-K workspace of {need} floats needed")
        d.ws = ws.data_ptr()
    d.no_reduce = 1 if (no_reduce and splitk > 1) else 0      # the consumer adds the slabs (single-phase launches)
    if splitk > 1 and not d.no_reduce and (FUSE_REDUCE if fuse_reduce is None else fuse_reduce):
        # one arrival counter per (phase, row tile, column tile), owned by this launch (two launches in flight on two streams must
        # not share them); zero now, and the last arriver of a tile leaves its counter at zero
        tickets = torch.zeros(d.nphase * (-(-(x.B * d.Hr * d.Wr) // 64)) * (wi.Npad // 64), dtype=torch.int32, device=x.t.device)
        d.tickets = tickets.data_ptr()
        keep.append(tickets)
    d.products = int(_PRODUCTS[-1] if products is None else products)
    d.k_order = int(wi.k_order)
    d.variant = int(variant)               # 0 / 2 = single-stage tiles, 4 = 64 x 128, 5 = pipelined, 6 = ping-pong (csrc/igemm.hip)
    return Launch(d, keep)


def splitk_for(M: int, Npad: int, ktiles: int, phases: int = 1, target: int = 768, phase_ktiles=None, bm: int = 128,
               min_ktiles: int = 16) -> int:
    """Split the reduction until the launch has about `target` workgroups (3 per CU: the LDS-DMA kernel's occupancy),
    keeping >= `min_ktiles` K tiles per slice (16; PWC-Net's 6x20 .. 24x80 grids with K of 30-150 tiles pass 4: a lone workgroup
    per CU pays ~1.1 us per K step, so eight slices of 8 steps beat two of 32 even with the reduction pass behind them).  `phase_ktiles` (K tiles of every phase) switches to a small cost model when the
    phases are unequal (stride-2 data gradients: 1, 2, 2 and 4 taps): `splitk` then counts the slices of the LONGEST
    phase, the others get proportionally fewer (ufr_igemm), and the choice minimises rounds x (slice length + epilogue)."""
    tiles_mn = -(-M // bm) * (Npad // (128 if Npad % 128 == 0 else 64))
    if phase_ktiles is not None and len(set(phase_ktiles)) > 1:
        best, best_cost = 1, None
        s = 1
        while s <= 16:
            per = -(-max(phase_ktiles) // s)
            if s > 1 and per < min(8, min_ktiles):
                break
            wgs = tiles_mn * sum(-(-k // per) for k in phase_ktiles)
            cost = -(-wgs // target) * (per + 6) + (10 if s > 1 else 0)     # K steps; epilogue ~ 6, the reduce launch ~ 10
            if best_cost is None or cost < best_cost:
                best, best_cost = s, cost
            s *= 2
        return best
    tiles = tiles_mn * phases
    s = 1
    while tiles * s * 2 <= target and ktiles // (s * 2) >= min_ktiles and s < 32:
        s *= 2
    return s

"""The adversarial-patch inner loop (patch_attacks/main.py:523-613) as a fused, graph-captured step.

Reference iteration (per `while loss_scalar > 0.1` pass):
    paste -> re-leaf -> predict_flow -> loss -> loss.backward() (weights included!) -> 9 elementwise
    kernels for `patch -= clamp(0.5*lr*(g_tgt+g_ref), -2, 2)` + re-paste + clamp -> loss.item() (host sync)
Here one iteration is
    forward (torch convs on MIOpen + gfx950 correlation) -> ufr_flow_loss (loss AND d loss/d flow)
    -> data-gradient-only backward -> ufr_patch_update (sum, step, clamp, re-paste, clamp: one kernel)
    -> ufr_attack_gate (the `loss > 0.1` / count bookkeeping, on the device)
captured once into a HIP graph and replayed; the host reads the 3-float gate state once per
attack() call instead of once per iteration.

Batch semantics (an extension: the reference is batch-1): B frame pairs (x N ranks) share ONE
patch in PATCH coordinates -- the object the reference carries from sample to sample is the canvas cropped at the
sample's placement (ry, rx) back to `patch_shape` (main.py:396-424).  The step holds P [1,3,ph,pw], its mask
[1,3,ph,pw] and a device-resident origin table [B,2]; pair b shows P at its own origin.  The loss is the mean over all
B*N*H*W pixels; per iteration every rank crops each pair's pre-clamp gradient at its placement, sums its pairs in
ascending order into [3,ph,pw] (31 KB at 51x51), all-gathers the N rows (+ the loss scalar in a 4-byte tail), and every
rank adds the rows in ascending rank order before `P -= clamp(0.5*lr*G, +-2)`: bit-identical patches on all ranks, a
patch pixel receives the gradient of every pair that shows it.  `sum_groups` splits a single process's pairs into the
same groups a sharded run has, which reproduces the N-rank summation tree bit for bit.
For ONE pair on one rank the reference's canvas-sized arithmetic is used as is (unmasked `g_tgt + g_ref`, main.py:581).
`shared_patch=False` runs B independent reference attacks (per-sample canvas patches, no exchange).

Cone of influence (cone.py, csrc/window.hip): only `mask * gradient` is ever used and only masked pixels
change between the iterations of one attack() call, so for networks that expose a convolutional prefix
(FlowNetC conv1-3 = 48% of its FLOPs) the prefix runs on a ~128x128 window around each pair's patch:
its adjoint every iteration, its forward from the second iteration on (the first full-frame prefix of a
call is computed once in load()).  Patch pixels outside the window -- never visible through the mask --
are left as loaded, where the reference adds image gradient to them.
"""
from __future__ import annotations

import ctypes as C
import os
from argparse import Namespace

import numpy as np
import torch
from ._lib import engine_cache as _engine_cache

from . import _lib as L
from .flownets.utils_model import predict_flow

LOSS_THRESHOLD = 0.1      # main.py:546
CLAMP_BOUND = 2.0         # main.py:581-583


def _pixel_range(flownet: str):
    """main.py:592-603: [0,1] for FlowNet*/RAFT/PWC, [-1,1] otherwise (SpyNet)."""
    return (0.0, 1.0) if any(k in flownet for k in ("FlowNetC", "FlowNetS", "FlowNet2", "RAFT", "PWC")) else (-1.0, 1.0)


class ShardedExchange:
    """The one exchange per iteration between the ranks that share a patch (RCCL: backend "nccl"; gloo in the CPU
    tests).  `gather`: all-gather of each rank's [groups, 3*ph*pw + 1] rows (cropped pre-clamp gradient sum | loss),
    issued before the non-linear update; every rank then adds the rows in the same order.  `__call__`: all-reduce(sum)
    of a packed buffer (the universal perturbation's image-sized gradient, universal_perturbation.py)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0

    def __call__(self, packed: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            self.dist.all_reduce(packed, op=self.dist.ReduceOp.SUM, group=self.group)
        return packed

    def gather(self, rows_local: torch.Tensor, rows_all: torch.Tensor) -> torch.Tensor:
        """rows_all[r*g:(r+1)*g] = rank r's rows_local ([g, n]); rows_all is [world*g, n], contiguous."""
        if self.world > 1:                       # one collective straight into the [world*g, n] buffer (no list of views)
            self.dist.all_gather_into_tensor(rows_all, rows_local, group=self.group)
        return rows_all


class PatchAttackStep:
    """Static buffers + the captured iteration for one (network, batch, resolution)."""

    def __init__(self, flow_net, args, batch, height, width, device="cuda:0", shared_patch=True,
                 exchange: ShardedExchange | None = None, use_graph=True, warmup=3, use_cone=None, patch_hw=None,
                 sum_groups=1):
        L.lib()   # fail loudly, now, if libufr_hip.so is missing
        self.net, self.args = flow_net, args
        self.B, self.H, self.W = batch, height, width
        self.dev = torch.device(device)
        self.shared = shared_patch
        self.exchange = exchange
        self.world = exchange.world if exchange is not None else 1
        if self.world > 1 and not shared_patch:
            raise ValueError("per-sample patches need no exchange; shard them as independent replicas")
        # several pairs behind one patch: the patch lives in patch coordinates (module docstring)
        self.placed = patch_hw is not None
        if shared_patch and batch * self.world > 1 and not self.placed:
            raise ValueError("one patch behind several pairs is defined in patch coordinates: pass patch_hw=(ph, pw) "
                             "and load() a [1,3,ph,pw] patch / mask with per-pair origins (SURVEY.md 8e)")
        if self.placed and not shared_patch:
            raise ValueError("patch_hw: per-sample patches stay canvas-sized like the reference's")
        self.groups = int(sum_groups)
        if self.groups < 1 or batch % self.groups or (self.groups > 1 and not self.placed):
            raise ValueError("sum_groups must divide the pairs of a patch-coordinate step")
        self.lo, self.hi = _pixel_range(args.flownet)
        self.kind = 1 if getattr(args, "l2", False) else 0
        self.alpha = float(getattr(args, "alpha", 0.0))
        self.step = 0.5 * float(args.lr)
        self.CHW = 3 * height * width
        f32 = dict(dtype=torch.float32, device=self.dev)
        pb = 1 if shared_patch else batch
        self.tgt = torch.zeros(batch, 3, height, width, **f32)
        self.ref = torch.zeros_like(self.tgt)
        self.mask = torch.zeros_like(self.tgt)            # canvas masks (placed mode: written by the placed paste)
        if self.placed:
            ph, pw = (int(v) for v in patch_hw)
            if not (0 < ph <= height and 0 < pw <= width):
                raise ValueError("patch_hw does not fit the frame")
            self.ph, self.pw, self.n_p = ph, pw, 3 * ph * pw
            self.patch = torch.zeros(1, 3, ph, pw, **f32)
            self.mask_p = torch.zeros(1, 3, ph, pw, **f32)
            self.origins = torch.zeros(batch, 2, dtype=torch.int32, device=self.dev)
            self.origins_host = None
            self.rows_local = torch.zeros(self.groups, self.n_p + 1, **f32)
            self.rows_all = torch.zeros(self.world * self.groups, self.n_p + 1, **f32) if self.world > 1 else self.rows_local
            self.loss_local = torch.zeros(1, **f32)
            self.loss_cur = torch.zeros(1, **f32)
        else:
            self.patch = torch.zeros(pb, 3, height, width, **f32)
            self.loss_cur = torch.zeros(1, **f32)
            self.loss_local = self.loss_cur
        self.patch_init = torch.zeros_like(self.patch)
        self.target = torch.zeros(batch, 2, height, width, **f32)
        self.adv_tgt = torch.zeros_like(self.tgt).requires_grad_(True)
        self.adv_ref = torch.zeros_like(self.tgt).requires_grad_(True)
        self.g_flow = torch.zeros_like(self.target)
        self.loss_ws = torch.zeros(L.LOSS_PARTIALS, **f32)     # workgroup partials of the fixed-order loss reduction
        # engine head: the loss kernel upsamples flow2 itself when its tiles fit the partial sums (else three passes)
        self._fused_loss = (batch * -(-(height // 4) // 16) * -(-(width // 4) // 16) <= L.LOSS_PARTIALS
                            and height % 4 == 0 and width % 4 == 0)
        self.g_flow2 = torch.zeros(batch, 2, height // 4, width // 4, **f32)
        self.state = torch.zeros(4, **f32)     # stopped, executed, last loss, (pad)
        # data gradient only: skips a third of the reference's FLOPs (main.py:573 `loss.backward()` also fills weight
        # gradients nothing reads).  The caller's flags are remembered: `release(flow_net)` puts them back.
        L.freeze_parameters(self.net)
        self.net.eval()
        self.graph = self.graph_b = self.graph_next = self.graph_b_next = None
        self._first = True
        self._rect_paste = False               # set while the "later iterations" form of the step is captured / run eagerly
        self.use_graph = use_graph
        self._warmup = warmup
        # windowed encoder (cone.py): networks that expose a convolutional prefix as CONE/encode/head
        if use_cone is None:
            use_cone = os.environ.get("UFR_CONE", "1") != "0"
        spec = getattr(flow_net, "CONE", None)
        self.cone = spec if (use_cone and spec is not None and height % spec.total_stride == 0
                             and width % spec.total_stride == 0) else None
        self.win_hw = None                     # static window size in pixels, fixed at the first load()
        self.win = torch.zeros(batch, 8, dtype=torch.int32, device=self.dev)     # one window per pair
        self.patch_loaded = torch.zeros_like(self.patch) if self.cone is not None else None

    # ------------------------------------------------------------------------------------ C ABI calls
    def _paste(self, do_clamp, gate=False, write_mask=False):
        if self.placed and self._rect_paste and not write_mask:
            # second and later iterations of a call: the canvas outside the patch rectangles already holds clamp(frame)
            L.check(L.lib().ufr_patch_paste_placed_rect(
                L.ptr(self.tgt), L.ptr(self.ref), L.ptr(self.patch), L.ptr(self.mask_p), L.ptr(self.origins),
                L.ptr(self.adv_tgt), L.ptr(self.adv_ref), self.B, self.H, self.W, self.ph, self.pw, int(do_clamp), self.lo,
                self.hi, L.ptr(self.state) if gate else None, L.stream()), "placed rect paste")
            return
        if self.placed:
            L.check(L.lib().ufr_patch_paste_placed(
                L.ptr(self.tgt), L.ptr(self.ref), L.ptr(self.patch), L.ptr(self.mask_p), L.ptr(self.origins),
                self.origins_host.ctypes.data if (write_mask and self.origins_host is not None) else None,
                L.ptr(self.adv_tgt), L.ptr(self.adv_ref), L.ptr(self.mask) if write_mask else None, self.B, self.H,
                self.W, self.ph, self.pw, int(do_clamp), self.lo, self.hi, L.ptr(self.state) if gate else None,
                L.stream()), "placed paste")
            return
        L.check(L.lib().ufr_patch_paste(L.ptr(self.tgt), L.ptr(self.ref), L.ptr(self.patch), L.ptr(self.mask),
                                        L.ptr(self.adv_tgt), L.ptr(self.adv_ref), self.B, self.CHW,
                                        0 if self.shared else self.CHW, self.CHW, int(do_clamp), self.lo,
                                        self.hi, L.stream()), "patch paste")

    def _update(self, g_tgt, g_ref):
        """Canvas form (one pair, or per-sample patches): the reference's arithmetic, main.py:575-600."""
        L.check(L.lib().ufr_patch_update(L.ptr(self.tgt), L.ptr(self.ref), L.ptr(g_tgt), L.ptr(g_ref),
                                         L.ptr(self.patch), L.ptr(self.mask), L.ptr(self.adv_tgt),
                                         L.ptr(self.adv_ref), self.B, self.CHW, 0 if self.shared else self.CHW,
                                         self.CHW, self.step, CLAMP_BOUND, self.lo, self.hi,
                                         L.ptr(self.state), L.stream()), "patch update")

    def _crop(self, g_tgt, g_ref):
        """Patch-coordinate form, local half: this rank's rows [groups, 3*ph*pw + 1] (cropped gradient sums | loss)."""
        L.check(L.lib().ufr_patch_grad_crop(L.ptr(g_tgt), L.ptr(g_ref), L.ptr(self.mask_p), L.ptr(self.origins), None,
                                            L.ptr(self.loss_local), L.ptr(self.rows_local), self.B, self.H, self.W,
                                            self.ph, self.pw, self.groups, L.stream()), "patch grad crop")

    def _apply(self):
        """Patch-coordinate form, common half (after the all-gather): fixed-order sum of every rank's rows, the
        clamped step on P, re-paste of every pair (main.py:581-600)."""
        L.check(L.lib().ufr_patch_apply(L.ptr(self.rows_all), self.rows_all.shape[0], L.ptr(self.patch),
                                        L.ptr(self.loss_cur), self.ph, self.pw, self.step, CLAMP_BOUND,
                                        L.ptr(self.state), L.stream()), "patch apply")
        self._paste(do_clamp=True, gate=True)

    # ------------------------------------------------------------------------------------ windowed encoder
    def _mask_extent(self):
        """Host read (once per window size): largest bounding-box extent of the loaded masks, over every
        rank so that all ranks size (and later re-size) their windows alike."""
        m = self.mask.amax(dim=1) != 0
        rows, cols = m.any(dim=2).cpu(), m.any(dim=1).cpu()
        ext = lambda v: max([int(r.nonzero().max() - r.nonzero().min()) + 1 if bool(r.any()) else 0 for r in v] + [1])
        eh, ew = ext(rows), ext(cols)
        if self.world > 1:
            both = torch.tensor([eh, ew], dtype=torch.float32, device=self.dev)
            self.exchange.dist.all_reduce(both, op=self.exchange.dist.ReduceOp.MAX, group=self.exchange.group)
            eh, ew = (int(v) for v in both.tolist())
        return eh, ew

    def _setup_cone(self):
        """Size the window for the loaded masks and allocate the window-sized / cached tensors; fall back
        to the full-frame iteration when the window would not be much smaller than the frame."""
        spec, B, H, W = self.cone, self.B, self.H, self.W
        eh, ew = self._mask_extent()
        wh, ww = spec.window_size(eh, H), spec.window_size(ew, W)
        self.graph = self.graph_b = self.graph_next = self.graph_b_next = None
        if self._warmup < 0:
            self._warmup = 1                   # re-capture after the window grew
        if wh * ww * 2 > H * W:
            self.cone = None
            return
        self.win_hw = (wh, ww)
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.xw = torch.zeros(2 * B, 3, wh, ww, **f32).requires_grad_(True)
        # native head (flownetc_engine.py / pwc_engine.py): the cached full-frame features live in its plane buffers, the head,
        # the windowed prefix and their adjoints are explicit launch schedules (no torch operator, no autograd between the
        # window stack and flow2)
        self.eng = self.eng_kind = None
        if getattr(self.net, "engine_available", None) is not None and self.net.engine_available(H, W, self.dev):
            self.eng_kind = getattr(self.net, "ENGINE", "flownetc")
            if self.eng_kind == "pwc":
                from .pwc_engine import get_engine
            else:
                from .flownetc_engine import get_engine
            self.eng = get_engine(self.net, B, H, W, self.dev)
            self.eng.flow_out.requires_grad_(True)
            self._wp_hold = self.eng.window_prefix(wh, ww)     # the captured graphs point into this state: keep it alive
        self.taps = []                         # (level stride, margin, frames, full leaf, window gradient)
        with torch.no_grad(), L.shape_probe():
            feats = self.net.encode(torch.zeros(2, 3, spec.total_stride * 2, spec.total_stride * 2, **f32))
        for t, m, fr, f in zip(spec.taps, spec.tap_margins(), spec.frames, feats):
            ls = spec.level_stride(t)
            n = B * fr                         # frames = 1: the head reads this tap for the first frame only
            full = None if self.eng is not None else torch.zeros(n, f.shape[1], H // ls, W // ls, **f32).requires_grad_(True)
            gwin = torch.zeros(n, f.shape[1], wh // ls, ww // ls, **f32)
            self.taps.append((ls, m, n, full, gwin))
        if not self.placed:                    # canvas-sized image gradients: only the reference's canvas form reads them
            self.g_tgt_full = torch.zeros_like(self.tgt)
            self.g_ref_full = torch.zeros_like(self.tgt)
        self._chain = spec.to_c()
        # column band for the head's most expensive data gradients (band_conv.py)
        self.band, reach = None, getattr(self.net, "BAND_REACH", None)
        if reach is not None and os.environ.get("UFR_BAND", "1") != "0":
            from .band_conv import Band
            bw = -(-(ww + 2 * reach + 31) // 32) * 32
            if W % 32 != 0 or bw * 4 > W * 3:
                bw = 0                         # frame too narrow for a band: only the correlation's adjoint is windowed
            self._band_reach = reach
            self.band = Band(torch.zeros(B, 8, dtype=torch.int32, device=self.dev), bw, cone_win=self.win,
                             cone_hw=(wh, ww))
            if bw and os.environ.get("UFR_INCREMENTAL", "1") != "0":
                self.band.inc_layers = tuple(getattr(self.net, "INCREMENTAL_LAYERS", ()))

    def _win_copy(self, fn, src, dst, n, c, hf, wf, ls, margin):
        wh, ww = self.win_hw
        L.check(fn(L.ptr(src), L.ptr(dst), L.ptr(self.win), self.B, n, c, hf, wf, wh // ls, ww // ls, ls, margin,
                   L.stream()), "window copy")

    def _cone_refresh(self, prefix_features=None):
        """New frames / new placement: window origin on the device, full prefix once (no autograd).
        `prefix_features` = `net.encode(cat(tgt, ref))` of the CLEAN frames, when the caller has just computed
        the clean forward anyway (train() does, for the target): outside the patch's cone they equal the
        adversarial frames' features and the first iteration's window pass overwrites the rest, so the
        full-frame prefix of this call is skipped."""
        lib = L.lib()
        wh, ww = self.win_hw
        L.check(lib.ufr_cone_window(L.ptr(self.mask), self.B, self.CHW, 3, self.H, self.W, C.byref(self._chain),
                                    wh, ww, L.ptr(self.win), L.ptr(self.state[3:]), L.stream()), "cone window")
        if not self.placed:                    # (a patch-coordinate step crops from the window gradients: no canvases)
            self.g_tgt_full.zero_(); self.g_ref_full.zero_()
        if self.band is not None and self.band.width:   # band start: 32-pixel aligned, `reach` left of the window, inside the frame
            start = torch.div(self.win[:, 1] - self._band_reach, 32, rounding_mode="floor") * 32
            self.band.win[:, 1] = start.clamp(0, self.W - self.band.width)
        if self.eng is not None:
            if prefix_features is not None:
                self.eng.load_prefix_features(*prefix_features)
            else:
                self.eng.prefix_full(self.adv_tgt.detach(), self.adv_ref.detach())
            return
        feats = prefix_features if prefix_features is not None else \
            self.net.encode(torch.cat((self.adv_tgt.detach(), self.adv_ref.detach()), 0))
        for (ls, m, n, full, _), f in zip(self.taps, feats):
            if f.shape[0] < n or f.shape[1:] != full.shape[1:]:
                raise ValueError("prefix_features do not match this step's frames")
            full.detach().copy_(f[:n])

    def _forward_cone(self):
        """Prefix on the window, paste into the cached full features, head at full size."""
        lib, B, H, W = L.lib(), self.B, self.H, self.W
        self._win_copy(lib.ufr_window_gather, self.adv_tgt, self.xw, B, 3, H, W, 1, 0)
        self._win_copy(lib.ufr_window_gather, self.adv_ref, self.xw[B:], B, 3, H, W, 1, 0)
        if self.eng is not None:               # the window's prefix on the engine too (no autograd graph)
            self._feats_w = None
            if self.eng_kind == "pwc":
                self.eng.window_prefix_forward(self.xw, self.win, self.taps[0][1])
                self.eng.forward_cached()
            else:
                (_, m2, _, _, _), (_, m3, _, _, _) = self.taps
                self.eng.window_prefix_forward(self.xw, self.win, m2, m3)
                self.eng.forward_cached(self.band)
            if self._fused_loss:                   # the loss kernel upsamples flow2 itself (no full-size flow)
                return None
            return torch.nn.functional.interpolate(self.eng.flow_out * self.eng.flow_scale, scale_factor=4, mode="bilinear",
                                                   align_corners=False)
        self._feats_w = [f[:n] for f, (_, _, n, _, _) in zip(self.net.encode(self.xw), self.taps)]
        for f, (ls, m, n, full, _) in zip(self._feats_w, self.taps):
            self._win_copy(lib.ufr_window_scatter, f, full, n, f.shape[1], H // ls, W // ls, ls, m)
        full = [t[3] for t in self.taps]
        extra = {"band": self.band} if self.band is not None else {}
        return self.net.head(*full[:-1], full[-1][:B], full[-1][B:], **extra)

    def _backward_cone(self, flow):
        """Adjoint of the head at full size, of the prefix on the window; canvas-sized gradients that are
        zero outside the window (only mask * gradient is ever used, main.py:575-583)."""
        lib, B, H, W = L.lib(), self.B, self.H, self.W
        if self.eng is not None:
            if flow is None:                     # ufr_flow2_upsampled_loss wrote d loss / d flow2 itself
                g_flow2 = self.g_flow2
            else:
                (g_flow2,) = torch.autograd.grad(flow, (self.eng.flow_out,), self.g_flow)
            if self.eng_kind == "pwc":           # one tap: level-2 features of both frames
                ls, m, n, _, gw = self.taps[0]
                g_f2 = self.eng.backward(g_flow2.contiguous())
                self._win_copy(lib.ufr_window_gather, g_f2, gw, n, g_f2.shape[1], H // ls, W // ls, ls, m)
                gxw = self.eng.window_prefix_backward(gw)
            else:
                (ls2, m2, _, _, gw2), (ls3, m3, _, _, gw3) = self.taps
                if self.band is not None:        # the engine writes conv3's window gradient itself (fused correlation adjoint)
                    self.band.g3_window, self.band.g3_margin = gw3, m3
                    self.band.eng_window, self.band.g2_margin = True, m2
                g2a, g3a, g3b = self.eng.backward(g_flow2.contiguous(), self.band)
                if g2a is not None:
                    self._win_copy(lib.ufr_window_gather, g2a, gw2, B, 128, H // ls2, W // ls2, ls2, m2)
                if g3a is not None:
                    self._win_copy(lib.ufr_window_gather, g3a, gw3, B, 256, H // ls3, W // ls3, ls3, m3)
                    self._win_copy(lib.ufr_window_gather, g3b, gw3[B:], B, 256, H // ls3, W // ls3, ls3, m3)
                gxw = self.eng.window_prefix_backward(self.taps[1][4], None if g2a is None else self.taps[0][4])
        else:
            g_full = torch.autograd.grad(flow, [t[3] for t in self.taps], self.g_flow)
            for g, (ls, m, n, _, gwin) in zip(g_full, self.taps):
                self._win_copy(lib.ufr_window_gather, g.contiguous(), gwin, n, g.shape[1], H // ls, W // ls, ls, m)
            gxw, = torch.autograd.grad(self._feats_w, (self.xw,), [t[4] for t in self.taps])
        self._feats_w = None
        gxw = gxw.contiguous()
        if self.placed:                        # the crop reads the window gradients directly (no canvas scatter, no canvas fill)
            return gxw, None
        self._win_copy(lib.ufr_window_scatter, gxw, self.g_tgt_full, B, 3, H, W, 1, 0)
        self._win_copy(lib.ufr_window_scatter, gxw[B:], self.g_ref_full, B, 3, H, W, 1, 0)
        return self.g_tgt_full, self.g_ref_full

    # ------------------------------------------------------------------------------------ one iteration
    def _part_a(self):
        """forward -> loss (+ d loss/d flow) -> data-gradient backward -> update (or, patch-coordinate form on
        several ranks: this rank's cropped gradient rows; the all-gather and `_part_b` follow)."""
        self.loss_local.zero_()
        if self.cone is not None:
            flow = self._forward_cone()
        else:
            flow = predict_flow(self.net, None, self.adv_tgt, self.adv_ref, self.args)
        # shared patch: loss = mean over the GLOBAL batch; private: every sample its own mean
        weight = (1.0 - self.alpha) * ((1.0 / self.world) if self.shared else float(self.B))
        if flow is None:        # engine head: loss on the x4-upsampled flow2 and its adjoint in one kernel (csrc/attack.hip)
            L.check(L.lib().ufr_flow2_upsampled_loss(L.ptr(self.eng.flow_out), float(self.eng.flow_scale), L.ptr(self.target),
                                                     L.ptr(self.g_flow2), L.ptr(self.loss_local), self.B, self.H // 4, self.W // 4,
                                                     self.kind, weight, L.ptr(self.loss_ws), L.stream()), "upsampled flow loss")
        else:
            if not flow.is_contiguous():
                flow = flow.contiguous()
            L.check(L.lib().ufr_flow_loss(L.ptr(flow), L.ptr(self.target), L.ptr(self.g_flow), L.ptr(self.loss_local),
                                          self.B, self.H * self.W, self.kind, weight, L.ptr(self.loss_ws), L.stream()),
                    "flow loss")
        if not self.shared:
            self.loss_local.div_(float(self.B))     # gate on the batch-mean loss
        if self.alpha != 0.0:                       # main.py:568-571 (scalar only: no gradient path)
            if self.placed:                         # the reference's mean runs over the canvas: same normaliser
                reg = (self.mask_p * self.patch - self.mask_p * self.patch_init).abs().sum() / float(self.CHW)
            else:
                reg = torch.nn.functional.l1_loss(self.mask * self.patch, self.mask * self.patch_init)
            self.loss_local.add_(self.alpha * reg / self.world)
        if self.cone is not None:
            g_tgt, g_ref = self._backward_cone(flow)
        else:
            g_tgt, g_ref = torch.autograd.grad(flow, (self.adv_tgt, self.adv_ref), self.g_flow)
            g_tgt, g_ref = g_tgt.contiguous(), g_ref.contiguous()
        if not self.placed:
            self._update(g_tgt, g_ref)
            self._gate()
            return
        if g_ref is None:                      # windowed prefix: crop from the window gradients [2B,3,wh,ww]
            wh, ww = self.win_hw
            L.check(L.lib().ufr_patch_grad_crop_window(L.ptr(g_tgt), L.ptr(self.win), L.ptr(self.mask_p), L.ptr(self.origins),
                                                       L.ptr(self.loss_local), L.ptr(self.rows_local), self.B, self.H, self.W, wh, ww,
                                                       self.ph, self.pw, self.groups, L.stream()), "patch grad crop (window)")
        else:
            self._crop(g_tgt, g_ref)
        if self.world == 1:
            self._apply()
            self._gate()

    def _part_b(self):
        """N>1 only, after the all-gather: the identical non-linear update on every rank."""
        self._apply()
        self._gate()

    def _gate(self):
        L.check(L.lib().ufr_attack_gate(L.ptr(self.loss_cur), L.ptr(self.state), LOSS_THRESHOLD, L.stream()),
                "attack gate")

    def _incremental_now(self):
        """From the second iteration after a load() the head's first blocks recompute the band only."""
        band = getattr(self, "band", None)
        return bool(self.cone is not None and band is not None and band.inc_layers and not self._first)

    def _set_later(self, later: bool):
        """The two forms of an iteration: the FIRST after a load() (full head forward, full-canvas clamped re-paste) and the
        LATER ones (band-only forward of the head's first blocks where the network has them; re-paste of the patch rectangles only)."""
        band = getattr(self, "band", None) if self.cone is not None else None
        if band is not None:
            band.incremental = bool(later and band.inc_layers)
        self._rect_paste = bool(later and self.placed and os.environ.get("UFR_RECT_PASTE", "1") != "0")

    def _iteration(self):
        later = not self._first
        if self.graph is not None:
            (self.graph_next if later else self.graph).replay()
        else:
            self._set_later(later)
            self._part_a()
        self._first = False
        if self.world > 1:
            self.exchange.gather(self.rows_local, self.rows_all)    # RCCL all-gather of [crop | loss] rows, eager, same stream
            if self.graph_b is not None:
                (self.graph_b_next if later else self.graph_b).replay()
            else:
                self._part_b()
        if self.graph is None:
            self._set_later(False)

    def _capture(self):
        """Warm up (MIOpen algorithm search, allocator) on a side stream, then capture the iteration in its two forms
        (`_set_later`): one graph each on a single GPU; two graphs each around the eager RCCL collective when sharded."""
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(self._warmup, 2)):      # at least one first and one later iteration
                self._iteration()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if not self.use_graph:
            return
        graphs = {}
        for later in (False, True):
            self._set_later(later)
            differs = later and (self._rect_paste or self._incremental_form())
            if later and not differs:
                graphs[True] = graphs[False]
                break
            ga = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ga):
                self._part_a()
            gb = None
            if self.world > 1:
                gb = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gb):
                    self._part_b()
            graphs[later] = (ga, gb)
        self._set_later(False)
        (self.graph, self.graph_b), (self.graph_next, self.graph_b_next) = graphs[False], graphs[True]

    def _incremental_form(self):
        band = getattr(self, "band", None) if self.cone is not None else None
        return bool(band is not None and band.incremental)

    # ------------------------------------------------------------------------------------ public API
    def load(self, tgt, ref, patch, mask, patch_init, target, prefix_features=None, origins=None):
        """Copy one attack() call's operands into the static buffers and do the first, un-clamped
        paste (main.py:537-542).  Must be called after capture (warm-up iterations move the patch).
        Patch-coordinate steps (`patch_hw`): patch / mask / patch_init are [1,3,ph,pw], `origins` [B,2] holds each
        pair's (row, column); canvas steps: canvas-sized tensors like the reference's.
        `prefix_features`: see `_cone_refresh` (ignored by the full-frame iteration)."""
        with torch.no_grad():
            self.tgt.copy_(tgt); self.ref.copy_(ref)
            if self.placed:
                if origins is None:
                    raise ValueError("a patch-coordinate step needs origins=[B,2] (row, column of every pair's patch)")
                if torch.is_tensor(origins):
                    self.origins.copy_(origins.reshape(self.B, 2))
                    self.origins_host = None              # device-resident placements: not validated on the host
                else:
                    oh = np.ascontiguousarray(np.asarray(origins, dtype=np.int32).reshape(self.B, 2))
                    if (oh < 0).any() or (oh[:, 0] + self.ph > self.H).any() or (oh[:, 1] + self.pw > self.W).any():
                        raise ValueError("a patch placement leaves the frame")
                    self.origins_host = oh
                    self.origins.copy_(torch.from_numpy(oh))
                self.mask_p.copy_(mask.reshape(self.mask_p.shape))
                self.patch.copy_(patch.reshape(self.patch.shape))
                self.patch_init.copy_(patch_init.reshape(self.patch.shape))
            else:
                self.mask.copy_(mask.expand_as(self.mask))
                self.patch.copy_(patch); self.patch_init.copy_(patch_init)
            self.target.copy_(target)
            self.state.zero_()
            self._paste(do_clamp=False, write_mask=True)
            self._first = True                 # the next iteration sees new frames: full head forward
            if self.cone is not None:
                self.patch_loaded.copy_(self.patch)
                if self.win_hw is None:
                    self._setup_cone()
                if self.cone is not None:
                    self._cone_refresh(prefix_features)

    def run(self, max_count):
        """Enqueue up to `max_count` iterations back to back; returns (executed, last_loss) after ONE
        host read of the device-side gate state."""
        with torch.cuda.device(self.dev):
            if self.graph is None and self._warmup >= 0:
                saved = [t.clone() for t in (self.tgt, self.ref, self.patch, self.mask_p if self.placed else self.mask,
                                             self.patch_init, self.target)]
                origins = (self.origins_host if self.origins_host is not None else self.origins.clone()) if self.placed else None
                self._capture()
                self._warmup = -1
                self.load(*saved, origins=origins)
            for _ in range(int(max_count)):
                self._iteration()
            if self.cone is not None and self.world > 1:      # a window overflow anywhere redoes the call everywhere
                self.exchange.dist.all_reduce(self.state[3:], op=self.exchange.dist.ReduceOp.MAX,
                                              group=self.exchange.group)
            st = self.state.tolist()
            if self.cone is not None and st[3] != 0.0:
                # a mask larger than the window this step was sized for: grow the window, re-capture, redo
                with torch.no_grad():
                    self.patch.copy_(self.patch_loaded)
                    self.state.zero_()
                    self._paste(do_clamp=False)
                    self._setup_cone()
                    if self.cone is not None:
                        self._cone_refresh()
                    self._first = True             # the cached head activations belong to the window that overflowed
                return self.run(max_count)
        return int(st[1]), float(st[2])

    def enqueue(self, iterations):
        """Throughput form: replay without reading anything back (bench.py)."""
        for _ in range(int(iterations)):
            self._iteration()


_STEP_CACHE_ATTR = "_ufr_patch_steps"


def release(flow_net):
    """Drop every cached step of `flow_net` (captured graphs, static buffers) and give its parameters back the
    `requires_grad` flags they had before the first step froze them -- e.g. before a training phase on the same module."""
    for attr in (_STEP_CACHE_ATTR, "_ufr_universal_steps"):
        cache = flow_net.__dict__.get(attr)
        if cache:
            cache.clear()
    # the caller's flags were recorded once per module at the first freeze (`_lib.freeze_parameters`), not per step: the step
    # cache is LRU, so neither its iteration order nor an eviction may decide what is restored
    L.restore_parameters(flow_net)


def attack(flow_net, tgt_img_var, ref_past_img_var, ref_future_img_var, patch_var, mask_var,
           patch_init_var, target_var, logger=None, args: Namespace | None = None, use_graph=True,
           prefix_features=None, origins=None):
    """Drop-in for patch_attacks/main.py::attack (:523-613): same positional arguments and return
    tuple `(adv_tgt, None, adv_ref_future, patch_var)`; `patch_var` is updated IN PLACE (:581).
    The reference reads the module-global `args` (main.py:534): when `args=` is not passed, the CALLER's module
    global `args` is used, so `from understanding_flow_robustness_amd.patch_attack import attack` inside
    patch_attacks/main.py works with the call at :396 unchanged (fields flownet, lr, alpha, l2, max_count).
    Extension: `origins=[B,2]` with [1,3,ph,pw] patch / mask tensors = one patch behind B pairs in patch coordinates."""
    if args is None:
        import sys
        frame = sys._getframe(1)
        args = frame.f_globals.get("args")
        if args is None or not hasattr(args, "flownet"):
            raise ValueError("attack(): no `args` Namespace in the calling module (main.py:534 reads a module global); "
                             "pass it as args=")
    L.require_hip(tgt_img_var, "tgt_img_var", contiguous=False)
    B, _, H, W = tgt_img_var.shape
    shared = patch_var.shape[0] == 1
    patch_hw = tuple(patch_var.shape[-2:]) if origins is not None else None
    key = (B, H, W, shared, patch_hw, bool(getattr(args, "l2", False)), float(args.lr), float(getattr(args, "alpha", 0.0)),
           args.flownet, bool(use_graph))
    cache = _engine_cache(flow_net, _STEP_CACHE_ATTR)
    step = cache.get(key)
    # a cached step holds captured graphs and (FlowNetC) pre-packed weights: a load_state_dict / in-place update of the
    # network since then makes it stale
    stamp = tuple((p.data_ptr(), p._version) for p in flow_net.parameters())
    if step is None or step.weights_stamp != stamp:
        step = cache[key] = PatchAttackStep(flow_net, args, B, H, W, device=tgt_img_var.device,
                                            shared_patch=shared, use_graph=use_graph, patch_hw=patch_hw)
        step.weights_stamp = stamp
    step.load(tgt_img_var, ref_future_img_var, patch_var, mask_var, patch_init_var, target_var,
              prefix_features=prefix_features, origins=origins)
    step.run(getattr(args, "max_count", 2))
    with torch.no_grad():
        patch_var.copy_(step.patch)
    return step.adv_tgt.detach().clone(), None, step.adv_ref.detach().clone(), patch_var


def _patch_type(args) -> str:
    """`--patch_type` (main.py:277-285): "circle" (the README's configuration) or "square"."""
    kind = getattr(args, "patch_type", "circle")
    if kind not in ("circle", "square"):
        raise ValueError("Please choose a square or circle patch")      # the reference's sys.exit message (main.py:285)
    return kind


def train_sample(flow_net, tgt_img, ref_past_img, ref_future_img, patch, mask, patch_init, patch_shape,
                 patch_shape_orig, args: Namespace, use_graph=True):
    """One loader item of patch_attacks/main.py::train (:363-461): clean forward, host-side
    `circle_transform` (numpy RNG consumed like the reference), H2D, the fused `attack`, D2H, crop at
    the placement and resample to the original patch size.  numpy patch state in, numpy patch state out:
    returns (patch, mask, patch_init, patch_shape).  Batch 1, like the reference (`patch[i]` indexing)."""
    from .utils_patch import circle_transform, crop_and_restore, square_transform
    dev = tgt_img.device
    with torch.no_grad():
        flow_pred = predict_flow(flow_net, ref_past_img, tgt_img, ref_future_img, args)
    if _patch_type(args) == "square":            # main.py:383-386 (no zoom: patch_shape stays)
        patch, mask, patch_init, rx, ry = square_transform(patch, mask, patch_init, tuple(tgt_img.shape), patch_shape,
                                                           norotate=getattr(args, "norotate", False))
    else:
        # NB the reference passes `True` as the 6th positional argument, i.e. margin=1 (main.py:377)
        patch, mask, patch_init, rx, ry, patch_shape = circle_transform(
            patch, mask, patch_init, tuple(tgt_img.shape), patch_shape, True)
    patch_t = torch.FloatTensor(patch).to(dev)
    mask_t = torch.FloatTensor(mask).to(dev)
    init_t = torch.FloatTensor(patch_init).to(dev)
    target = -flow_pred.detach()
    _, _, _, patch_t = attack(flow_net, tgt_img, ref_past_img, ref_future_img, patch_t, mask_t, init_t, target,
                              None, args=args, use_graph=use_graph)
    masked = torch.mul(mask_t, patch_t)
    return crop_and_restore(masked.cpu().numpy(), mask_t.cpu().numpy(), init_t.cpu().numpy(), rx, ry, patch_shape,
                            patch_shape_orig)


def train_sample_device(flow_net, tgt_img, ref_past_img, ref_future_img, patch, mask, patch_init, patch_shape,
                        patch_shape_orig, args: Namespace, use_graph=True):
    """`train_sample` with the patch state resident on the device (float64 HIP tensors [1,3,S,S], SURVEY.md 8
    f2): placement, attack, crop and resampling all run on the GPU (patch_transform.py); the only host work is
    drawing the reference's `np.random` numbers.  Returns (patch, mask, patch_init, patch_shape) as HIP tensors."""
    from .patch_transform import circle_transform_device, crop_and_restore_device, square_transform_device
    with torch.no_grad():
        prefix = None
        if (getattr(flow_net, "CONE", None) is not None and not flow_net.training
                and os.environ.get("UFR_SEED_PREFIX", "1") != "0"):
            # the clean forward in two halves: its conv1-3 features seed the attack's cache (no second
            # full-frame prefix in load()); same operations as flow_net(tgt, ref)
            B = tgt_img.shape[0]
            prefix = flow_net.encode(torch.cat((tgt_img, ref_future_img), 0))
            flow_pred = flow_net.head(*[f[:B] for f in prefix[:-1]], prefix[-1][:B], prefix[-1][B:])
        else:
            flow_pred = predict_flow(flow_net, ref_past_img, tgt_img, ref_future_img, args)
        if _patch_type(args) == "square":
            patch_t, mask_t, init_t, rx, ry, _ = square_transform_device(patch, mask, patch_init, tuple(tgt_img.shape), patch_shape,
                                                                         norotate=getattr(args, "norotate", False))
        else:
            patch_t, mask_t, init_t, rx, ry, patch_shape = circle_transform_device(
                patch, mask, patch_init, tuple(tgt_img.shape), patch_shape, True)     # margin=1, main.py:377
    target = -flow_pred.detach()
    _, _, _, patch_t = attack(flow_net, tgt_img, ref_past_img, ref_future_img, patch_t, mask_t, init_t, target,
                              None, args=args, use_graph=use_graph, prefix_features=prefix)
    with torch.no_grad():
        return crop_and_restore_device(patch_t, mask_t, init_t, rx, ry, patch_shape, patch_shape_orig)


ERROR_NAMES = ["epe", "adv_epe", "cos_sim", "adv_cos_sim"]


def validate_flow_with_gt(patch, mask, patch_shape, val_loader, flow_net, args: Namespace):
    """patch_attacks/main.py::validate_flow_with_gt (:616-784): for every validation item place the
    patch (`circle_transform`, same RNG order), run the clean and the patched pair, and average
    EPE / cosine similarity against the ground-truth flow.  Returns `(errors_avg, error_names)`.
    `patch` / `mask` may be the reference's numpy arrays or float64 HIP tensors.

    Differences from the reference, none of them numeric: the placement runs on the device
    (patch_transform.py: no canvas-sized host arrays, no H2D per item), the clean and the adversarial pair
    run as ONE batch of two, paste+clamp is the fused kernel, the four metrics stay on the device and the
    host synchronises once at the end instead of four times per item (`.item()` in losses.py)."""
    from . import losses
    from .patch_transform import circle_transform_device, square_transform_device
    square = _patch_type(args) == "square"
    flow_net.eval()
    lo, hi = _pixel_range(args.flownet)
    sums, count, patch_d, mask_d = None, 0, None, None
    with torch.no_grad():
        for item in val_loader:
            ref_past, tgt, ref_future, flow_gt = item[0], item[1], item[2], item[3]
            dev = tgt.device
            L.require_hip(tgt, "tgt_img", contiguous=False)
            if patch_d is None:
                f64 = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a))).to(dev, torch.float64)
                patch_d, mask_d = f64(patch), f64(mask)
            if square:                           # main.py:646-654; the turns accumulate (written back to the caller's arrays below)
                patch_t, mask_t, _, _, _, (patch_d, mask_d, _) = square_transform_device(
                    patch_d, mask_d, patch_d, tuple(tgt.shape), patch_shape, norotate=getattr(args, "norotate", False))
            else:
                patch_t, mask_t, _, _, _, _ = circle_transform_device(patch_d, mask_d, patch_d, tuple(tgt.shape), patch_shape)
            tgt, ref_future = tgt.contiguous(), ref_future.contiguous()
            B, _, H, W = tgt.shape
            adv_tgt, adv_ref = torch.empty_like(tgt), torch.empty_like(ref_future)
            L.check(L.lib().ufr_patch_paste(L.ptr(tgt), L.ptr(ref_future), L.ptr(patch_t), L.ptr(mask_t), L.ptr(adv_tgt),
                                            L.ptr(adv_ref), B, 3 * H * W, 3 * H * W, 3 * H * W, 1, lo, hi, L.stream()),
                    "patch paste")
            flows = predict_flow(flow_net, None, torch.cat((tgt, adv_tgt)), torch.cat((ref_future, adv_ref)), args)
            flow_fwd, adv_flow = flows[:B], flows[B:]
            vals = torch.stack([losses.epe_tensor(flow_gt, flow_fwd), losses.epe_tensor(flow_gt, adv_flow),
                                losses.cossim_tensor(flow_gt, flow_fwd), losses.cossim_tensor(flow_gt, adv_flow)])
            sums = vals if sums is None else sums + vals
            count += 1
    if square and patch_d is not None:
        # the reference's square_transform turns the CALLER's patch and mask in place (utils_patch.py:793-798, called with the
        # training state at main.py:647): the accumulated turns are part of the state the next epoch starts from
        for dst, src in ((patch, patch_d), (mask, mask_d)):
            if torch.is_tensor(dst):
                dst.copy_(src.to(dst.dtype))
            else:
                np.copyto(dst, src.cpu().numpy().astype(dst.dtype, copy=False))
    avg = (sums / max(count, 1)).tolist() if sums is not None else [0.0] * 4
    return avg, list(ERROR_NAMES)

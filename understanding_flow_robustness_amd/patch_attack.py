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

Batch semantics (an extension: the reference is batch-1): B frame pairs share ONE
canvas-sized patch; the loss is the mean over all B*H*W pixels, so the update uses
sum_b d(loss)/d(adv_b) -- for B = 1 exactly the reference.  With N ranks each holds B/N pairs; the
pre-clamp gradient sum (+ the loss) is all-reduced over RCCL before the non-linear update, so every
rank applies the identical update (`ShardedExchange`).
"""
from __future__ import annotations

import ctypes as C
from argparse import Namespace

import torch

from . import _lib as L
from .flownets.utils_model import predict_flow

LOSS_THRESHOLD = 0.1      # main.py:546
CLAMP_BOUND = 2.0         # main.py:581-583


def _pixel_range(flownet: str):
    """main.py:592-603: [0,1] for FlowNet*/RAFT/PWC, [-1,1] otherwise (SpyNet)."""
    return (0.0, 1.0) if any(k in flownet for k in ("FlowNetC", "FlowNetS", "FlowNet2", "RAFT", "PWC")) else (-1.0, 1.0)


class ShardedExchange:
    """All-reduce(sum) of the packed [pre-clamp gradient sum | loss] buffer across the ranks that
    share the patch.  One collective per iteration, issued before the non-linear update."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, packed: torch.Tensor) -> torch.Tensor:
        if self.world > 1:
            self.dist.all_reduce(packed, op=self.dist.ReduceOp.SUM, group=self.group)
        return packed


class PatchAttackStep:
    """Static buffers + the captured iteration for one (network, batch, resolution)."""

    def __init__(self, flow_net, args, batch, height, width, device="cuda:0", shared_patch=True,
                 exchange: ShardedExchange | None = None, use_graph=True, warmup=3):
        L.lib()   # fail loudly, now, if libufr_hip.so is missing
        self.net, self.args = flow_net, args
        self.B, self.H, self.W = batch, height, width
        self.dev = torch.device(device)
        self.shared = shared_patch
        self.exchange = exchange
        self.world = exchange.world if exchange is not None else 1
        if self.world > 1 and not shared_patch:
            raise ValueError("per-sample patches need no exchange; shard them as independent replicas")
        self.lo, self.hi = _pixel_range(args.flownet)
        self.kind = 1 if getattr(args, "l2", False) else 0
        self.alpha = float(getattr(args, "alpha", 0.0))
        self.step = 0.5 * float(args.lr)
        self.CHW = 3 * height * width
        f32 = dict(dtype=torch.float32, device=self.dev)
        pb = 1 if shared_patch else batch
        self.tgt = torch.zeros(batch, 3, height, width, **f32)
        self.ref = torch.zeros_like(self.tgt)
        self.mask = torch.zeros_like(self.tgt)
        self.patch = torch.zeros(pb, 3, height, width, **f32)
        self.patch_init = torch.zeros_like(self.patch)
        self.target = torch.zeros(batch, 2, height, width, **f32)
        self.adv_tgt = torch.zeros_like(self.tgt).requires_grad_(True)
        self.adv_ref = torch.zeros_like(self.tgt).requires_grad_(True)
        self.g_flow = torch.zeros_like(self.target)
        # packed exchange buffer: [3*H*W pre-clamp gradient sum | loss of this iteration]
        self.packed = torch.zeros(self.CHW + 1, **f32)
        self.loss_cur = self.packed[self.CHW:]
        self.state = torch.zeros(4, **f32)     # stopped, executed, last loss, (pad)
        for p in self.net.parameters():        # data gradient only: skips a third of the reference's FLOPs
            p.requires_grad_(False)
        self.net.eval()
        self.graph = self.graph_b = None
        self.use_graph = use_graph
        self._warmup = warmup

    # ------------------------------------------------------------------------------------ C ABI calls
    def _paste(self, do_clamp):
        L.check(L.lib().ufr_patch_paste(L.ptr(self.tgt), L.ptr(self.ref), L.ptr(self.patch), L.ptr(self.mask),
                                        L.ptr(self.adv_tgt), L.ptr(self.adv_ref), self.B, self.CHW,
                                        0 if self.shared else self.CHW, self.CHW, int(do_clamp), self.lo,
                                        self.hi, L.stream()), "patch paste")

    def _update(self, g_tgt, g_ref, mode):
        L.check(L.lib().ufr_patch_update(L.ptr(self.tgt), L.ptr(self.ref), L.ptr(g_tgt) if g_tgt is not None else None,
                                         L.ptr(g_ref) if g_ref is not None else None, L.ptr(self.packed),
                                         L.ptr(self.patch), L.ptr(self.mask), L.ptr(self.adv_tgt),
                                         L.ptr(self.adv_ref), self.B, self.CHW, 0 if self.shared else self.CHW,
                                         self.CHW, self.step, CLAMP_BOUND, self.lo, self.hi, mode,
                                         L.ptr(self.state), L.stream()), "patch update")

    # ------------------------------------------------------------------------------------ one iteration
    def _part_a(self):
        """forward -> loss (+ d loss/d flow) -> data-gradient backward -> [N>1: local gradient sum]."""
        self.loss_cur.zero_()
        flow = predict_flow(self.net, None, self.adv_tgt, self.adv_ref, self.args)
        if not flow.is_contiguous():
            flow = flow.contiguous()
        # shared patch: loss = mean over the GLOBAL batch; private: every sample its own mean
        weight = (1.0 - self.alpha) * ((1.0 / self.world) if self.shared else float(self.B))
        L.check(L.lib().ufr_flow_loss(L.ptr(flow), L.ptr(self.target), L.ptr(self.g_flow), L.ptr(self.loss_cur),
                                      self.B, self.H * self.W, self.kind, weight, L.stream()), "flow loss")
        if not self.shared:
            self.loss_cur.div_(float(self.B))       # gate on the batch-mean loss
        if self.alpha != 0.0:                       # main.py:568-571 (scalar only: no gradient path)
            reg = torch.nn.functional.l1_loss(self.mask * self.patch, self.mask * self.patch_init)
            self.loss_cur.add_(self.alpha * reg / self.world)
        g_tgt, g_ref = torch.autograd.grad(flow, (self.adv_tgt, self.adv_ref), self.g_flow)
        g_tgt, g_ref = g_tgt.contiguous(), g_ref.contiguous()
        if self.world > 1:
            self._update(g_tgt, g_ref, 1)            # local sum -> packed; the collective follows
        else:
            self._update(g_tgt, g_ref, 0)
            self._gate()

    def _part_b(self):
        """N>1 only, after the all-reduce: the identical non-linear update on every rank."""
        self._update(None, None, 2)
        self._gate()

    def _gate(self):
        L.check(L.lib().ufr_attack_gate(L.ptr(self.loss_cur), L.ptr(self.state), LOSS_THRESHOLD, L.stream()),
                "attack gate")

    def _iteration(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._part_a()
        if self.world > 1:
            self.exchange(self.packed)               # RCCL all-reduce of [grad sum | loss], eager, same stream
            if self.graph_b is not None:
                self.graph_b.replay()
            else:
                self._part_b()

    def _capture(self):
        """Warm up (MIOpen algorithm search, allocator) on a side stream, then capture the iteration:
        one graph on a single GPU; two graphs around the eager RCCL collective when sharded."""
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(self._warmup, 1)):
                self._iteration()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if not self.use_graph:
            return
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self._part_a()
        graph_b = None
        if self.world > 1:
            graph_b = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_b):
                self._part_b()
        self.graph, self.graph_b = graph, graph_b

    # ------------------------------------------------------------------------------------ public API
    def load(self, tgt, ref, patch, mask, patch_init, target):
        """Copy one attack() call's operands into the static buffers and do the first, un-clamped
        paste (main.py:537-542).  Must be called after capture (warm-up iterations move the patch)."""
        with torch.no_grad():
            self.tgt.copy_(tgt); self.ref.copy_(ref); self.mask.copy_(mask.expand_as(self.mask))
            self.patch.copy_(patch); self.patch_init.copy_(patch_init); self.target.copy_(target)
            self.state.zero_()
            self._paste(do_clamp=False)

    def run(self, max_count):
        """Enqueue up to `max_count` iterations back to back; returns (executed, last_loss) after ONE
        host read of the device-side gate state."""
        with torch.cuda.device(self.dev):
            if self.graph is None and self._warmup >= 0:
                saved = [t.clone() for t in (self.tgt, self.ref, self.mask, self.patch, self.patch_init, self.target)]
                self._capture()
                self._warmup = -1
                self.load(*[saved[i] for i in (0, 1, 3, 2, 4, 5)])
            for _ in range(int(max_count)):
                self._iteration()
            st = self.state.tolist()
        return int(st[1]), float(st[2])

    def enqueue(self, iterations):
        """Throughput form: replay without reading anything back (bench.py)."""
        for _ in range(int(iterations)):
            self._iteration()


_STEP_CACHE_ATTR = "_ufr_patch_steps"


def attack(flow_net, tgt_img_var, ref_past_img_var, ref_future_img_var, patch_var, mask_var,
           patch_init_var, target_var, logger=None, args: Namespace | None = None, use_graph=True):
    """Drop-in for patch_attacks/main.py::attack (:523-613): same positional arguments and return
    tuple `(adv_tgt, None, adv_ref_future, patch_var)`; `patch_var` is updated IN PLACE (:581).
    The reference reads the module-global `args`; pass it as `args=` (fields flownet, lr, alpha, l2,
    max_count)."""
    if args is None:
        raise ValueError("attack(): pass the CLI Namespace as args= (the reference reads a module global)")
    L.require_hip(tgt_img_var, "tgt_img_var", contiguous=False)
    B, _, H, W = tgt_img_var.shape
    shared = patch_var.shape[0] == 1
    key = (B, H, W, shared, bool(getattr(args, "l2", False)), float(args.lr), float(getattr(args, "alpha", 0.0)),
           args.flownet, bool(use_graph))
    cache = flow_net.__dict__.setdefault(_STEP_CACHE_ATTR, {})
    step = cache.get(key)
    if step is None:
        step = cache[key] = PatchAttackStep(flow_net, args, B, H, W, device=tgt_img_var.device,
                                            shared_patch=shared, use_graph=use_graph)
    step.load(tgt_img_var, ref_future_img_var, patch_var, mask_var, patch_init_var, target_var)
    step.run(getattr(args, "max_count", 2))
    with torch.no_grad():
        patch_var.copy_(step.patch)
    return step.adv_tgt.detach().clone(), None, step.adv_ref.detach().clone(), patch_var


def train_sample(flow_net, tgt_img, ref_past_img, ref_future_img, patch, mask, patch_init, patch_shape,
                 patch_shape_orig, args: Namespace, use_graph=True):
    """One loader item of patch_attacks/main.py::train (:363-461): clean forward, host-side
    `circle_transform` (numpy RNG consumed like the reference), H2D, the fused `attack`, D2H, crop at
    the placement and resample to the original patch size.  numpy patch state in, numpy patch state out:
    returns (patch, mask, patch_init, patch_shape).  Batch 1, like the reference (`patch[i]` indexing)."""
    from .utils_patch import circle_transform, crop_and_restore
    dev = tgt_img.device
    with torch.no_grad():
        flow_pred = predict_flow(flow_net, ref_past_img, tgt_img, ref_future_img, args)
    if getattr(args, "patch_type", "circle") != "circle":
        raise NotImplementedError("only --patch_type circle (the README's configuration) is mirrored")
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


ERROR_NAMES = ["epe", "adv_epe", "cos_sim", "adv_cos_sim"]


def validate_flow_with_gt(patch, mask, patch_shape, val_loader, flow_net, args: Namespace):
    """patch_attacks/main.py::validate_flow_with_gt (:616-784): for every validation item place the
    patch (host `circle_transform`, same RNG order), run the clean and the patched pair, and average
    EPE / cosine similarity against the ground-truth flow.  Returns `(errors_avg, error_names)`.

    Differences from the reference, none of them numeric: the clean and the adversarial pair run as
    ONE batch of two, paste+clamp is the fused kernel, the four metrics stay on the device and the
    host synchronises once at the end instead of four times per item (`.item()` in losses.py)."""
    from . import losses
    from .utils_patch import circle_transform
    if getattr(args, "patch_type", "circle") != "circle":
        raise NotImplementedError("only --patch_type circle is mirrored")
    flow_net.eval()
    lo, hi = _pixel_range(args.flownet)
    sums, count = None, 0
    with torch.no_grad():
        for item in val_loader:
            ref_past, tgt, ref_future, flow_gt = item[0], item[1], item[2], item[3]
            dev = tgt.device
            L.require_hip(tgt, "tgt_img", contiguous=False)
            patch_full, mask_full, _, _, _, _ = circle_transform(patch, mask, patch.copy(), tuple(tgt.shape), patch_shape)
            patch_t = torch.FloatTensor(patch_full).to(dev)
            mask_t = torch.FloatTensor(mask_full).to(dev)
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
    avg = (sums / max(count, 1)).tolist() if sums is not None else [0.0] * 4
    return avg, list(ERROR_NAMES)

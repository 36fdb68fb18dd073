"""Universal-perturbation / I-FGSM inner loop (global_attacks/universal_perturbation.py:452-530,
:667-675; loss from global_attacks/perturb_model.py:102-145) as a fused, graph-captured step.

Reference step (x n_step, default 10):  re-leaf both frames -> predict_flow -> compute_flow_loss ->
backward(retain_graph=True, weights included) -> sign -> lr*sign -> adv -/+ -> clamp[0,1] ->
noise = clamp(adv - img, +-output_norm) -> adv = img + noise.
Here: forward -> ufr_flow_loss_ex (loss + d loss/d flow) -> data-gradient backward ->
ufr_universal_update (everything after the backward, both frames, one kernel); one HIP graph per step.

Batch semantics: with B = 1 and no exchange the arithmetic is the reference's, sample by sample
(`shared=False`, delta [B,2,3,H,W]).  `shared=True` is the build's extension for a sharded batch
(BASELINE config 5): ONE perturbation [2,3,H,W]; the step direction is the sign of the gradient
SUMMED over all samples of all ranks (all-reduced before the sign), so every rank applies the
identical update.  For B = 1 its adversarial frames equal the reference's wherever the [0,1]
image-range clamp does not bind (clamping to [0,1] then to img+-eps is clamping to the
intersection); where it binds, the reference folds the clamp into its per-sample noise while the
shared delta stays image-independent (tests/test_models_cpu.py pins both statements).
"""
from __future__ import annotations

from argparse import Namespace

import torch
from ._lib import engine_cache as _engine_cache

from . import _lib as L
from .flownets.utils_model import predict_flow
from .patch_attack import ShardedExchange

_KINDS = {"cossim": 0, "l2": 1, "l1": 2}
_FRAMES = {"both": 3, "left": 1, "right": 2}


def add_universal_perturbation(image0, image1, universal_perturbation, lower_bound=0.0, upper_bound=1.0):
    """universal_perturbation.py:667-675, including its asymmetric indexing ([0,0] vs [:,1])."""
    if universal_perturbation.shape[1] != 2:
        raise Exception("Universarial perturbation: first dimension must be 2!")
    return (torch.clamp(image0 + universal_perturbation[0, 0], lower_bound, upper_bound),
            torch.clamp(image1 + universal_perturbation[:, 1], lower_bound, upper_bound))


class UniversalPerturbationStep:
    def __init__(self, model, args, batch, height, width, gt_channels=2, device="cuda:0", shared=False,
                 exchange: ShardedExchange | None = None, use_graph=True, warmup=2):
        L.lib()
        self.model, self.args = model, args
        self.B, self.H, self.W, self.Cg = batch, height, width, gt_channels
        self.dev = torch.device(device)
        self.shared = shared
        self.exchange = exchange
        self.world = exchange.world if exchange is not None else 1
        if self.world > 1 and not shared:
            raise ValueError("a sharded batch needs the shared perturbation")
        if args.flow_loss not in _KINDS:
            raise NotImplementedError(f"flow_loss {args.flow_loss!r}")
        method = args.perturb_method.lower()
        if "ifgsm" in method:
            self.use_sign = 1
        elif method == "ifgm":
            self.use_sign = 0
        else:
            raise NotImplementedError(method)
        self.kind = _KINDS[args.flow_loss]
        self.frames = _FRAMES[args.perturb_mode]
        self.lr, self.eps = float(args.learning_rate), float(args.output_norm)
        self.ascent = 1 if getattr(args, "add_gaussian", False) else 0
        self.CHW = 3 * height * width
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.img0 = torch.zeros(batch, 3, height, width, **f32)
        self.img1 = torch.zeros_like(self.img0)
        self.gt = torch.zeros(batch, gt_channels, height, width, **f32)
        self.adv0 = torch.zeros_like(self.img0).requires_grad_(True)
        self.adv1 = torch.zeros_like(self.img0).requires_grad_(True)
        self.g_flow = torch.zeros(batch, 2, height, width, **f32)
        self.delta = torch.zeros((2, 3, height, width) if shared else (batch, 2, 3, height, width), **f32)
        self.packed = torch.zeros(2 * self.CHW + 1, **f32)      # [grad sum frame 0 | frame 1 | loss]
        self.loss_cur = self.packed[2 * self.CHW:]
        self.scale_t = torch.ones(1, **f32)     # 1/normaliser of the loss, device-resident
        self.loss_ws = torch.zeros(L.LOSS_PARTIALS, **f32)     # workgroup partials of the fixed-order loss reduction
        L.freeze_parameters(self.model)                         # patch_attack.release() restores the caller's flags
        self.model.eval()
        self.graph = self.graph_b = None
        self.use_graph, self._warmup, self._captured = use_graph, warmup, False

    def _update(self, g0, g1, mode):
        L.check(L.lib().ufr_universal_update(
            L.ptr(self.img0), L.ptr(self.img1), L.ptr(g0) if g0 is not None else None,
            L.ptr(g1) if g1 is not None else None, L.ptr(self.packed), L.ptr(self.adv0), L.ptr(self.adv1),
            L.ptr(self.delta), self.B, self.CHW, self.lr, self.eps, 0.0, 1.0, self.use_sign, self.frames,
            self.ascent, int(self.shared), mode, L.stream()), "universal update")

    def _part_a(self):
        self.loss_cur.zero_()
        flow = predict_flow(self.model, None, self.adv0, self.adv1, self.args).contiguous()
        L.check(L.lib().ufr_flow_loss_ex(L.ptr(flow), L.ptr(self.gt), L.ptr(self.g_flow), L.ptr(self.loss_cur), self.B,
                                         self.H * self.W, self.Cg, self.kind, 0.0, L.ptr(self.scale_t), L.ptr(self.loss_ws),
                                         L.stream()), "flow loss")
        g0, g1 = torch.autograd.grad(flow, (self.adv0, self.adv1), self.g_flow, allow_unused=True)
        g0 = torch.zeros_like(self.img0) if g0 is None else g0.contiguous()
        g1 = torch.zeros_like(self.img1) if g1 is None else g1.contiguous()   # :479-483
        self._update(g0, g1, 1 if self.world > 1 else 0)

    def _iteration(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self._part_a()
        if self.world > 1:
            self.exchange(self.packed)
            if self.graph_b is not None:
                self.graph_b.replay()
            else:
                self._update(None, None, 2)

    def _capture(self):
        side = torch.cuda.Stream(device=self.dev)
        side.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(side):
            for _ in range(max(self._warmup, 1)):
                self._iteration()
        torch.cuda.current_stream(self.dev).wait_stream(side)
        torch.cuda.synchronize(self.dev)
        if self.use_graph:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self._part_a()
            graph_b = None
            if self.world > 1:
                graph_b = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph_b):
                    self._update(None, None, 2)
            self.graph, self.graph_b = graph, graph_b
        self._captured = True

    def load(self, img0, img1, universal_perturbation, target):
        """Operands of one attack() call + the initial add_universal_perturbation (:461-463)."""
        with torch.no_grad():
            self.img0.copy_(img0); self.img1.copy_(img1); self.gt.copy_(target)
            if self.shared:
                self.delta.copy_(universal_perturbation.reshape(-1, 2, 3, self.H, self.W)[0])
                a0 = torch.clamp(self.img0 + self.delta[0], 0.0, 1.0)
                a1 = torch.clamp(self.img1 + self.delta[1], 0.0, 1.0)
            else:
                a0, a1 = add_universal_perturbation(self.img0, self.img1, universal_perturbation)
            self.adv0.copy_(a0); self.adv1.copy_(a1)
            if self.Cg == 3:      # 1 / (sum(valid) + eps), perturb_model.py:141-143, computed on the device
                valid = self.gt[:, 2].sum().reshape(1)
                if self.world > 1:
                    self.exchange.dist.all_reduce(valid)
                self.scale_t.copy_(1.0 / (valid + 1e-8))
            else:
                per_pix = 2 if self.kind == 2 else 1
                self.scale_t.fill_(1.0 / (self.B * self.world * self.H * self.W * per_pix))

    def run(self, n_step):
        with torch.cuda.device(self.dev):
            if not self._captured:
                saved = (self.img0.clone(), self.img1.clone(), self.delta.clone(), self.gt.clone(),
                         self.adv0.detach().clone(), self.adv1.detach().clone())
                self._capture()
                with torch.no_grad():
                    self.img0.copy_(saved[0]); self.img1.copy_(saved[1]); self.delta.copy_(saved[2])
                    self.gt.copy_(saved[3]); self.adv0.copy_(saved[4]); self.adv1.copy_(saved[5])
            for _ in range(int(n_step)):
                self._iteration()

    enqueue = run


_CACHE = "_ufr_universal_steps"


def attack(model, img0_var, img1_var, universal_perturbation_var, target_var, args: Namespace, use_graph=True):
    """Drop-in for global_attacks/universal_perturbation.py::attack (:452-530):
    returns (adv_img0, None, adv_img1, universal_perturbation [B,2,3,H,W])."""
    L.require_hip(img0_var, "img0_var", contiguous=False)
    B, _, H, W = img0_var.shape
    key = (B, H, W, target_var.shape[1], args.flow_loss, args.perturb_method, args.perturb_mode,
           float(args.learning_rate), float(args.output_norm), bool(getattr(args, "add_gaussian", False)),
           args.flownet, bool(use_graph))
    cache = _engine_cache(model, _CACHE)
    step = cache.get(key)
    if step is None:
        step = cache[key] = UniversalPerturbationStep(model, args, B, H, W, gt_channels=target_var.shape[1],
                                                      device=img0_var.device, shared=False, use_graph=use_graph)
    step.load(img0_var, img1_var, universal_perturbation_var, target_var)
    step.run(args.n_step)
    return step.adv0.detach().clone(), None, step.adv1.detach().clone(), step.delta.detach().clone()

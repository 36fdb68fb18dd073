"""Per-image global attacks of global_attacks/perturb_model.py for optical flow (`disparity=False`):
`compute_flow_loss` (:102-145) and `PerturbationsModel` with FGSM (:423-473), I-FGSM / I-FGM
(:475-619), MI-FGSM (:621-757) and the "none" method; input diversity (:759-821) runs eagerly.

The iterative methods reuse the fused universal-perturbation step (universal_perturbation.py): the
reference's I-FGSM is the same arithmetic with gradient ASCENT (`image + lr*sign(grad)`, :557-571),
descent when `targeted`.  The two noise methods (uniform :332-382, gaussian :274-330) draw on the host like
the reference and upload once; `imagecorruptions` methods are CPU image-library code and out of the hot
path's scope.
"""
from __future__ import annotations

import random
from argparse import Namespace

import numpy as np
import torch
from ._lib import engine_cache as _engine_cache
import torch.nn as nn

from . import _lib as L
from .flownets.utils_model import predict_flow
from .universal_perturbation import UniversalPerturbationStep


def compute_flow_loss(flow_net, image0, image1, ground_truth, args):
    """perturb_model.py:102-145 (plain torch: used by the eager paths and by callers that want the scalar)."""
    epsilon = 1e-8
    flow_output = predict_flow(flow_net, None, image0, image1, args)
    if args.flow_loss == "cossim":
        loss = 1 - nn.functional.cosine_similarity(flow_output, ground_truth[:, :2, ...])
    elif args.flow_loss == "l2":
        loss = (torch.sum((flow_output - ground_truth[:, :2, ...]) ** 2, dim=1) + 10e-8).sqrt()
    elif args.flow_loss == "l1":
        loss = (flow_output - ground_truth[:, :2, ...]).abs()
    else:
        raise NotImplementedError
    if ground_truth.shape[1] == 3:
        valid = ground_truth[:, 2, ...]
        return (loss * valid).sum() / (valid.sum() + epsilon)
    return loss.mean()


def adversarial_training_batch(model, image1, image2, flow, valid, gt_full, args, arbitrary_gt=None):
    """The adversarial-training step of training/train.py:171-222 up to the optimiser: attack the current
    batch with `PerturbationsModel` (model in eval mode), measure the EPE of the attacked prediction, and
    return the clean + adversarial batch for fine-tuning:
        (image1, image2, flow, valid, epe_attacked)   with batch size doubled.
    `arbitrary_gt` = (flow, valid) of another sample turns the attack into the targeted one (:188-198).
    The fused attack steps freeze the parameters while they run; their `requires_grad` flags and the
    module's train/eval mode are restored before returning, so the caller's optimiser step works."""
    from .flownets.utils_model import predict_flow as _predict
    from .losses import compute_epe
    flags = [p.requires_grad for p in model.parameters()]
    was_training = model.training
    model.eval()
    try:
        pm = PerturbationsModel(perturb_method=args.perturb_method, perturb_mode=args.perturb_mode,
                                output_norm=args.output_norm, n_step=args.perturb_n_step,
                                learning_rate=args.perturb_learning_rate, momentum=args.perturb_momentum,
                                probability_diverse_input=args.probability_diverse_input, disparity=False,
                                targeted=arbitrary_gt is not None, args=args)
        if arbitrary_gt is not None:
            perturb_gt = torch.cat((arbitrary_gt[0], arbitrary_gt[1][:, None, ...]), dim=1)
        else:
            perturb_gt = torch.cat((flow, valid[:, None, ...]), dim=1)
        _, _, image1_adv, image2_adv = pm.forward(model, image1, image2, perturb_gt)
        with torch.no_grad():
            flow_output = _predict(model, None, image1_adv, image2_adv, args)
        epe_attacked = compute_epe(gt=gt_full, pred=flow_output)
    finally:
        for p, f in zip(model.parameters(), flags):
            p.requires_grad_(f)
        model.__dict__.pop(L._GRAD_FLAGS_ATTR, None)          # the caller's flags are back: a later freeze records afresh
        model.train(was_training)
    return (torch.cat((image1, image1_adv)), torch.cat((image2, image2_adv)), torch.cat((flow, flow)),
            torch.cat((valid, valid)), epe_attacked)


class PerturbationsModel:
    """perturb_model.py:148-272: same constructor vocabulary and `forward` contract
    `(noise0, noise1, image0_adv, image1_adv)`."""

    def __init__(self, perturb_method="fgsm", perturb_mode="both", output_norm=0.02, n_step=40, learning_rate=2e-3,
                 momentum=0.47, probability_diverse_input=0.0, device=torch.device("cuda"), disparity=False,
                 targeted=False, show_perturbation_evolution=None, print_out=False, args=None, use_graph=True):
        if disparity:
            raise NotImplementedError("stereo (disparity) models are not part of this repository's flow path")
        if show_perturbation_evolution:
            raise NotImplementedError("GIF export is visualisation, out of scope")
        self.method, self.mode = perturb_method, perturb_mode
        self.eps, self.n_step, self.lr, self.mu = float(output_norm), int(n_step), float(learning_rate), float(momentum)
        self.p_diverse = float(probability_diverse_input)
        self.targeted, self.args, self.use_graph = bool(targeted), args, use_graph
        self._steps = {}

    # ------------------------------------------------------------------------------------------ helpers
    def _grads(self, model, image0, image1, ground_truth):
        image0 = image0.detach().requires_grad_(True)
        image1 = image1.detach().requires_grad_(True)
        i0, i1, gt = self._diverse_input(image0, image1, ground_truth)
        loss = compute_flow_loss(model, i0, i1, gt, self.args)
        if self.targeted:
            loss = loss * -1
        g0, g1 = torch.autograd.grad(loss, (image0, image1), allow_unused=True)
        return g0, (torch.zeros_like(image1) if g1 is None else g1)

    def _diverse_input(self, image0, image1, ground_truth):
        """perturb_model.py:759-821: random down-scale (>= 90%) + zero pad back, flow rescaled."""
        if torch.rand(1) > self.p_diverse:
            return image0, image1, ground_truth
        _, _, oh, ow = image0.shape
        nh, nw = random.randint(int(oh - oh / 10.0), oh), random.randint(int(ow - ow / 10.0), ow)
        top = random.randint(0, oh - nh)
        left = random.randint(0, ow - nw)
        pad = (left, ow - nw - left, top, oh - nh - top)
        rs = lambda t, mode: nn.functional.pad(nn.functional.interpolate(t, size=(nh, nw), mode=mode), pad=pad,
                                               mode="constant", value=0)
        gt = rs(ground_truth, "nearest") * (float(nw) / float(ow))
        return rs(image0, "bilinear"), rs(image1, "bilinear"), gt

    def _mode_mask(self, n0, n1):
        if self.mode == "both":
            return n0, n1
        if self.mode == "left":
            return n0, torch.zeros_like(n1)
        if self.mode == "right":
            return torch.zeros_like(n0), n1
        raise ValueError("Invalid perturbation mode: %s" % self.mode)

    # ------------------------------------------------------------------------------------------ methods
    def _fgsm(self, model, image0, image1, ground_truth):
        g0, g1 = self._grads(model, image0, image1, ground_truth)          # :442-456
        return self._mode_mask(self.eps * torch.sign(g0), self.eps * torch.sign(g1))

    def _iterative_step(self, model, image0, ground_truth):
        B, _, H, W = image0.shape
        # cached on the network (a new PerturbationsModel per training batch, train.py:172-186, must not
        # re-capture the graph): weights are read through their storage, so optimiser updates are seen
        self._steps = _engine_cache(model, "_ufr_perturb_steps")
        key = (B, H, W, ground_truth.shape[1], self.method, self.mode, self.lr, self.eps, self.targeted, self.use_graph,
               getattr(self.args, "flow_loss", None), getattr(self.args, "flownet", None))
        step = self._steps.get(key)
        if step is None:
            sargs = Namespace(**vars(self.args))
            sargs.perturb_method = "ifgsm" if self.method in ("ifgsm", "mifgsm") else "ifgm"
            sargs.perturb_mode, sargs.learning_rate, sargs.output_norm = self.mode, self.lr, self.eps
            sargs.add_gaussian = not self.targeted                        # ascent on the loss unless targeted
            step = self._steps[key] = UniversalPerturbationStep(model, sargs, B, H, W, gt_channels=ground_truth.shape[1],
                                                                device=image0.device, shared=False,
                                                                use_graph=self.use_graph)
        return step

    def _ifgsm(self, model, image0, image1, ground_truth):
        if self.p_diverse > 0.0:
            return self._iterative_eager(model, image0, image1, ground_truth, momentum=False)
        step = self._iterative_step(model, image0, ground_truth)
        zero = torch.zeros(image0.shape[0], 2, *image0.shape[1:], device=image0.device)
        step.load(image0, image1, zero, ground_truth)
        step.run(self.n_step)
        return step.delta[:, 0].clone(), step.delta[:, 1].clone()

    def _iterative_eager(self, model, image0, image1, ground_truth, momentum):
        """:475-619 / :621-757 step by step (input diversity draws host RNG every step; MI-FGSM keeps an
        L1-normalised running gradient); the update itself is still the fused kernel."""
        B, _, H, W = image0.shape
        CHW = 3 * H * W
        adv0, adv1 = image0.clone().contiguous(), image1.clone().contiguous()
        i0c, i1c = image0.contiguous(), image1.contiguous()
        delta = torch.zeros(B, 2, 3, H, W, device=image0.device)
        m0, m1 = torch.zeros_like(image0), torch.zeros_like(image1)
        frames = {"both": 3, "left": 1, "right": 2}[self.mode]
        use_sign = 0 if self.method == "ifgm" else 1       # MI-FGSM always steps along sign(momentum) (:657-668)
        for _ in range(self.n_step):
            g0, g1 = self._grads(model, adv0, adv1, ground_truth)
            if momentum:                                                   # :651-656
                m0 = self.mu * m0 + (1.0 - self.mu) * g0 / torch.sum(torch.abs(g0))
                m1 = self.mu * m1 + (1.0 - self.mu) * g1 / torch.sum(torch.abs(g1))
                g0, g1 = m0, m1
            g0, g1 = g0.contiguous(), g1.contiguous()
            L.check(L.lib().ufr_universal_update(L.ptr(i0c), L.ptr(i1c), L.ptr(g0), L.ptr(g1), None, L.ptr(adv0),
                                                 L.ptr(adv1), L.ptr(delta), B, CHW, self.lr, self.eps, 0.0, 1.0, use_sign,
                                                 frames, 1, 0, 0, L.stream()), "iterative update")
        return delta[:, 0].clone(), delta[:, 1].clone()

    def _uniform(self, image0, image1):
        """:332-382: two `np.random.uniform` draws (left frame first) whatever the mode, float64 -> float32."""
        draws = [np.random.uniform(size=tuple(im.shape), low=-self.eps, high=self.eps) for im in (image0, image1)]
        n0, n1 = (torch.from_numpy(d).float().to(image0.device) for d in draws)
        return self._mode_mask(n0, n1)

    def _gaussian(self, image0, image1):
        """:274-330 through `skimage.util.random_noise(mode="gaussian", var=(eps/4)^2)`, restated (skimage is not
        in this image): image + N(0, var) in float64, clipped to [0,1] ([-1,1] for an image with negative values),
        cast to float32; the noise is the difference to the input.  skimage >= 0.19 draws from an unseeded
        `default_rng()`, so no stream is reproducible; this one follows the `np.random` global state."""
        sigma = self.eps / 4.0
        out = []
        for im in (image0, image1):
            x = im.detach().cpu().numpy()
            low = -1.0 if x.min() < 0 else 0.0
            noisy = np.clip(x + np.random.normal(0.0, sigma, x.shape), low, 1.0)
            out.append(torch.from_numpy(noisy).float().to(im.device) - im)
        return self._mode_mask(*out)

    # ------------------------------------------------------------------------------------------ API
    def forward(self, model, image0, image1, ground_truth):
        """perturb_model.py:211-272."""
        L.require_hip(image0, "image0", contiguous=False)
        if self.method in ("fgsm", "fgm"):
            noise0, noise1 = self._fgsm(model, image0, image1, ground_truth)
        elif self.method in ("ifgsm", "ifgm"):
            noise0, noise1 = self._ifgsm(model, image0, image1, ground_truth)
        elif self.method in ("mifgsm", "mifgm"):
            noise0, noise1 = self._iterative_eager(model, image0, image1, ground_truth, momentum=True)
        elif self.method == "gaussian":
            noise0, noise1 = self._gaussian(image0, image1)
        elif self.method == "uniform":
            noise0, noise1 = self._uniform(image0, image1)
        elif self.method == "none":
            noise0, noise1 = torch.zeros_like(image0), torch.zeros_like(image1)
        else:
            raise NotImplementedError(f"perturbation method {self.method!r} (imagecorruptions) is out of scope")
        image0_output = torch.clamp(image0 + noise0, 0.0, 1.0)
        image1_output = torch.clamp(image1 + noise1, 0.0, 1.0)
        return image0_output - image0, image1_output - image1, image0_output, image1_output

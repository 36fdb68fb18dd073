"""CPU restatement of the model-level hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Functional (state_dict in, tensors out) torch-fp32 restatements of the reference's flow networks and
attack loops, with the native operators routed to the C oracle (oracle/oracle_ops.py).  Pinned by
golden fixtures generated from the reference's own Python modules (tests/golden/make_golden_models.py).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this file.

Each function cites the reference file:line it follows.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import oracle_ops as oo

_RGB_MEAN = torch.tensor([0.40066648, 0.39482617, 0.3784785], dtype=torch.float64).view(1, 3, 1, 1)


def _lrelu(x):
    return F.leaky_relu(x, 0.1)


def _conv(sd, name, x, stride=1, act=True):
    w = sd[name + ".weight"]
    y = F.conv2d(x, w, sd.get(name + ".bias"), stride=stride, padding=(w.shape[-1] - 1) // 2)
    return _lrelu(y) if act else y


def _deconv(sd, name, x, act=True):
    y = F.conv_transpose2d(x, sd[name + ".weight"], sd.get(name + ".bias"), stride=2, padding=1)
    return _lrelu(y) if act else y


def correlate(a, b, patch=21, dil_patch=2):
    """models/submodules.py:124-138."""
    out = oo.spatial_correlation_sample(a.contiguous(), b.contiguous(), kernel_size=1, patch_size=patch,
                                        stride=1, padding=0, dilation_patch=dil_patch)
    bsz, ph, pw, h, w = out.shape
    return out.view(bsz, ph * pw, h, w) / a.size(1)


def flownetc_forward(sd, x1, x2, div_flow=20.0):
    """models/FlowNetC.py:81-197 (eval branch)."""
    x1 = (x1.double() - _RGB_MEAN).float()                      # :73-79,:93-94
    x2 = (x2.double() - _RGB_MEAN).float()
    c1a = _conv(sd, "conv1.0", x1, 2); c2a = _conv(sd, "conv2.0", c1a, 2); c3a = _conv(sd, "conv3.0", c2a, 2)
    c1b = _conv(sd, "conv1.0", x2, 2); c2b = _conv(sd, "conv2.0", c1b, 2); c3b = _conv(sd, "conv3.0", c2b, 2)
    corr = _lrelu(correlate(c3a, c3b))                          # :134,:139
    redir = _conv(sd, "conv_redir.0", c3a)                      # :142
    c3_1 = _conv(sd, "conv3_1.0", torch.cat((redir, corr), 1))
    c4 = _conv(sd, "conv4_1.0", _conv(sd, "conv4.0", c3_1, 2))
    c5 = _conv(sd, "conv5_1.0", _conv(sd, "conv5.0", c4, 2))
    c6 = _conv(sd, "conv6_1.0", _conv(sd, "conv6.0", c5, 2))
    flow6 = _conv(sd, "predict_flow6", c6, act=False)
    cat5 = torch.cat((c5, _deconv(sd, "deconv5.0", c6), _deconv(sd, "upsampled_flow6_to_5", flow6, act=False)), 1)
    flow5 = _conv(sd, "predict_flow5", cat5, act=False)
    cat4 = torch.cat((c4, _deconv(sd, "deconv4.0", cat5), _deconv(sd, "upsampled_flow5_to_4", flow5, act=False)), 1)
    flow4 = _conv(sd, "predict_flow4", cat4, act=False)
    cat3 = torch.cat((c3_1, _deconv(sd, "deconv3.0", cat4), _deconv(sd, "upsampled_flow4_to_3", flow4, act=False)), 1)
    flow3 = _conv(sd, "predict_flow3", cat3, act=False)
    cat2 = torch.cat((c2a, _deconv(sd, "deconv2.0", cat3), _deconv(sd, "upsampled_flow3_to_2", flow3, act=False)), 1)
    flow2 = _conv(sd, "predict_flow2", cat2, act=False)
    return F.interpolate(flow2 * div_flow, scale_factor=4, mode="bilinear", align_corners=False)  # :194-197


# ------------------------------------------------------------------------------------------- attack
def flow_loss(flow, target, l2=False):
    """patch_attacks/main.py:557-566."""
    if l2:
        return (torch.sum((flow - target) ** 2, dim=1) + 1e-8).sqrt().mean()
    return (1 - F.cosine_similarity(flow, target)).mean()


def patch_attack(predict, tgt, ref, patch, mask, patch_init, target, lr=1e3, alpha=0.0, max_count=2,
                 l2=False, clamp=(0.0, 1.0), trace=None):
    """patch_attacks/main.py:523-613 for a `predict(adv_tgt, adv_ref) -> flow` callable.

    `patch` is updated in place like the reference's patch_var; returns
    (adv_tgt, adv_ref, patch, executed_iterations, last_loss).  With a batch of B > 1 the patch is
    shared ([1,3,H,W]) and the per-sample gradients are summed before the clamp (the build's batch
    extension); B = 1 is the reference's arithmetic exactly.
    """
    adv_tgt = (1 - mask) * tgt + mask * patch                   # :537-542
    adv_ref = (1 - mask) * ref + mask * patch
    count, loss_scalar = 0, 1.0
    while loss_scalar > 0.1:                                    # :546
        count += 1
        adv_tgt = adv_tgt.detach().requires_grad_(True)
        adv_ref = adv_ref.detach().requires_grad_(True)
        flow = predict(adv_tgt, adv_ref)
        loss_data = flow_loss(flow, target, l2)
        loss_reg = F.l1_loss(mask * patch, mask * patch_init)   # :568-570 (no gradient path to the update)
        loss = (1 - alpha) * loss_data + alpha * loss_reg
        g_tgt, g_ref = torch.autograd.grad(loss, (adv_tgt, adv_ref))
        g = g_tgt + g_ref
        if patch.shape[0] == 1 and g.shape[0] > 1:
            g = g.sum(0, keepdim=True)
        patch -= torch.clamp(0.5 * lr * g, -2, 2)               # :581-583
        adv_tgt = torch.clamp((1 - mask) * tgt + mask * patch, *clamp)   # :585-600
        adv_ref = torch.clamp((1 - mask) * ref + mask * patch, *clamp)
        loss_scalar = float(loss)                               # :605
        if trace is not None:
            trace.append(dict(loss=loss_scalar, patch=patch.clone(), adv_tgt=adv_tgt.detach().clone(),
                              adv_ref=adv_ref.detach().clone(), g_tgt=g_tgt.clone(), g_ref=g_ref.clone()))
        if count > max_count - 1:                               # :610-611
            break
    return adv_tgt.detach(), adv_ref.detach(), patch, count, loss_scalar


# ------------------------------------------------------------------------------------------- metrics
def compute_epe(gt, pred):
    """patch_attacks/losses.py:8-28."""
    _, _, h_pred, w_pred = pred.size()
    bs, nc, h_gt, w_gt = gt.size()
    pred = F.interpolate(pred, size=(h_gt, w_gt), mode="bilinear", align_corners=False)
    u = pred[:, 0] * (w_gt / w_pred)
    v = pred[:, 1] * (h_gt / h_pred)
    epe = torch.sqrt((gt[:, 0] - u) ** 2 + (gt[:, 1] - v) ** 2)
    if nc == 3:
        valid = gt[:, 2]
        return float((epe * valid).sum() / (valid.sum() + 1e-8))
    return float(epe.sum() / (bs * h_gt * w_gt))


def compute_cossim(gt, pred):
    """patch_attacks/losses.py:31-50."""
    bs, nc, h_gt, w_gt = gt.size()
    pred = F.interpolate(pred, size=(h_gt, w_gt), mode="bilinear", align_corners=False)
    sim = F.cosine_similarity(gt[:, :2], pred)
    if nc == 3:
        valid = gt[:, 2]
        return float((sim * valid).sum() / (valid.sum() + 1e-8))
    return float(sim.sum() / (bs * h_gt * w_gt))

"""EPE / cosine-similarity metrics (patch_attacks/losses.py:8-50): bilinear resize of the prediction
to the ground truth's size, u/v rescale, optional validity mask (3rd GT channel).  Returned as
Python floats like the reference (one host sync per call)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

epsilon = 1e-8


def _resize(pred, gt):
    return F.interpolate(pred, size=gt.shape[-2:], mode="bilinear", align_corners=False)


def epe_tensor(gt, pred):
    """compute_epe without the host read: a 0-d tensor (batched validation keeps it on the device)."""
    _, _, h_pred, w_pred = pred.size()
    bs, nc, h_gt, w_gt = gt.size()
    pred = _resize(pred, gt)
    u_pred = pred[:, 0] * (w_gt / w_pred)
    v_pred = pred[:, 1] * (h_gt / h_pred)
    epe = torch.sqrt(torch.pow(gt[:, 0] - u_pred, 2) + torch.pow(gt[:, 1] - v_pred, 2))
    if nc == 3:
        valid = gt[:, 2]
        return (epe * valid).sum() / (valid.sum() + epsilon)
    return epe.sum() / (bs * h_gt * w_gt)


def cossim_tensor(gt, pred):
    bs, nc, h_gt, w_gt = gt.size()
    pred = _resize(pred, gt)
    similarity = F.cosine_similarity(gt[:, :2], pred)
    if nc == 3:
        valid = gt[:, 2]
        return (similarity * valid).sum() / (valid.sum() + epsilon)
    return similarity.sum() / (bs * h_gt * w_gt)


def compute_epe(gt, pred):
    return epe_tensor(gt, pred).item()


def compute_cossim(gt, pred):
    return cossim_tensor(gt, pred).item()


def multiscale_cossim(gt, pred):
    assert len(gt) == len(pred)
    return sum(-F.cosine_similarity(g, p).mean() for g, p in zip(gt, pred))

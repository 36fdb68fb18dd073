"""End-point error and cosine similarity against a ground-truth flow (the metrics of
patch_attacks/losses.py:8-50).  The prediction is brought to the ground truth's size bilinearly and its
components are rescaled by the size ratio; a third ground-truth channel, when present, is a validity mask
and the mean runs over valid pixels only.  `*_tensor` keep the result on the device (batched validation
reads once at the end); `compute_*` return Python floats like the reference."""
from __future__ import annotations

import torch
import torch.nn.functional as F

epsilon = 1e-8


def _at_gt_size(gt, pred):
    return F.interpolate(pred, size=gt.shape[-2:], mode="bilinear", align_corners=False)


def _mean_over_valid(per_pixel, gt):
    """Mean of a [B,H,W] map: over the pixels flagged in gt[:, 2] when there is such a channel, else over all."""
    if gt.shape[1] == 3:
        valid = gt[:, 2]
        return (per_pixel * valid).sum() / (valid.sum() + epsilon)
    return per_pixel.sum() / (gt.shape[0] * gt.shape[2] * gt.shape[3])


def epe_tensor(gt, pred):
    scale_u, scale_v = gt.shape[3] / pred.shape[3], gt.shape[2] / pred.shape[2]
    pred = _at_gt_size(gt, pred)
    du, dv = gt[:, 0] - pred[:, 0] * scale_u, gt[:, 1] - pred[:, 1] * scale_v
    return _mean_over_valid(torch.sqrt(torch.pow(du, 2) + torch.pow(dv, 2)), gt)


def cossim_tensor(gt, pred):
    return _mean_over_valid(F.cosine_similarity(gt[:, :2], _at_gt_size(gt, pred)), gt)


def compute_epe(gt, pred):
    return epe_tensor(gt, pred).item()


def compute_cossim(gt, pred):
    return cossim_tensor(gt, pred).item()


def multiscale_cossim(gt, pred):
    """Negative mean cosine similarity summed over the scales of a flow pyramid."""
    assert len(gt) == len(pred)
    return sum(-F.cosine_similarity(g, p).mean() for g, p in zip(gt, pred))

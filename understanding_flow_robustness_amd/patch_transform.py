"""The patch's trip between its own SxS frame and the image canvas, on the device.

Device counterparts of `utils_patch.circle_transform` (patch_attacks/utils_patch.py:257-358) and of the
crop / zoom-back after the attack (patch_attacks/main.py:408-461).  The patch state is a float64 HIP tensor
[1,3,S,S] -- the reference's numpy arrays are float64 too -- and never visits the host: the reference spends
~80 ms per sample on three canvas-sized `np.zeros`, `scipy.ndimage` calls, float64->float32 conversion and
three H2D copies (plus the D2H on the way back), several times the attack iteration itself.

RNG mapping (bit-documented): `np.random` is consumed on the host in exactly the reference's order --
`random()` brightness offset, `random()` zoom factor, then per sample `random()` rotation angle,
`choice(W - 2S - 2 margin - 2)` column, `choice(H - 2S - 2)` row -- and the drawn numbers are passed to the
kernels; the resampling itself (csrc/placement.hip) reproduces scipy.ndimage's order-0/1, mode='constant'
arithmetic in float64.
"""
from __future__ import annotations

import numpy as np
import torch
from scipy import special

from . import _lib as L


def _resample(src, out_hw, m, off, order):
    """src [N,C,h,w] float64 HIP tensor -> [N,C,oh,ow]; planes share the 2x2 matrix `m` and `off`."""
    L.require_hip(src, "patch state", contiguous=True)
    if src.dtype != torch.float64:
        raise TypeError("the patch state is float64 (like the reference's numpy arrays)")
    n, c, h, w = src.shape
    dst = torch.empty(n, c, out_hw[0], out_hw[1], dtype=torch.float64, device=src.device)
    L.check(L.lib().ufr_affine_resample_f64(L.ptr(src), L.ptr(dst), n * c, h, w, out_hw[0], out_hw[1], float(m[0]),
                                            float(m[1]), float(m[2]), float(m[3]), float(off[0]), float(off[1]),
                                            int(order), L.stream()), "affine resample")
    return dst


def zoom_device(x, factors, order):
    """scipy.ndimage.zoom(x, (1, 1, fy, fx), order=order) (mode='constant', grid_mode=False)."""
    h, w = x.shape[-2:]
    oh, ow = int(round(h * factors[0])), int(round(w * factors[1]))
    zy = (h - 1) / (oh - 1) if oh > 1 else 1.0
    zx = (w - 1) / (ow - 1) if ow > 1 else 1.0
    return _resample(x, (oh, ow), (zy, 0.0, 0.0, zx), (0.0, 0.0), order)


def rotate_device(x, angle, order=1):
    """scipy.ndimage.rotate(plane, angle, reshape=False, order=order) on every [h,w] plane of x."""
    h, w = x.shape[-2:]
    c, s = special.cosdg(angle), special.sindg(angle)
    rot = np.array([[c, s], [-s, c]])
    shape = np.asarray((h, w))
    out_center = rot @ ((shape - 1) / 2)
    offset = (shape - 1) / 2 - out_center
    return _resample(x, (h, w), (rot[0, 0], rot[0, 1], rot[1, 0], rot[1, 1]), offset, order)


def circle_transform_device(patch, mask, patch_init, data_shape, patch_shape, margin=0, center=False, norotate=False,
                            fixed_loc=(-1, -1), moving=False):
    """utils_patch.circle_transform with HIP tensors: float64 state in, float32 canvases [1,3,H,W] out.
    Returns (canvas_patch, canvas_mask, canvas_init, x, y, zoomed patch shape)."""
    if data_shape[0] != 1 or patch.shape[0] != 1:
        raise NotImplementedError("batch 1, like the reference (`patch[i]` for i < data_shape[0])")
    if not moving:
        patch = patch + np.random.random() * 0.1 - 0.05      # two array operations, like the reference
    patch = torch.clamp(patch, 0.0, 1.0) * mask
    image_w, image_h = data_shape[-1], data_shape[-2]
    if not moving:
        f = 1 + 0.05 * (np.random.random() - 0.5)
        patch = zoom_device(patch.contiguous(), (f, f), 1)
        mask = zoom_device(mask.contiguous(), (f, f), 0)
        patch_init = zoom_device(patch_init.contiguous(), (f, f), 1)
    new_shape = tuple(patch.shape)
    side = patch.shape[-1]
    if not norotate:
        angle = 10 * (np.random.random() - 0.5)
        patch, patch_init = rotate_device(patch.contiguous(), angle), rotate_device(patch_init.contiguous(), angle)
    if fixed_loc[0] < 0 or fixed_loc[1] < 0:
        if center:
            x, y = (image_w - side) // 2, (image_h - side) // 2
        else:
            x = side + margin + np.random.choice(image_w - 2 * side - 2 * margin - 2)
            y = side + np.random.choice(image_h - 2 * side - 2)
        assert x + side < image_w and y + side < image_h
    else:
        x, y = fixed_loc
    C = patch.shape[1]
    canvases = [torch.empty(1, C, image_h, image_w, dtype=torch.float32, device=patch.device) for _ in range(3)]
    L.check(L.lib().ufr_patch_place(L.ptr(patch.contiguous()), L.ptr(mask.contiguous()), L.ptr(patch_init.contiguous()), C,
                                    new_shape[-2], new_shape[-1], L.ptr(canvases[0]), L.ptr(canvases[1]),
                                    L.ptr(canvases[2]), image_h, image_w, int(y), int(x), L.stream()), "patch place")
    return canvases[0], canvases[1], canvases[2], int(x), int(y), new_shape


def square_transform_device(patch, mask, patch_init, data_shape, patch_shape, norotate=False):
    """utils_patch.square_transform (utils_patch.py:781-846) with HIP tensors: float64 state in, float32 canvases out.
    RNG: `choice(4)` quarter turns (unless norotate), `choice(W - S - 1)`, `choice(H - S - 1)`.  The reference rotates
    the caller's arrays in place; tensors are immutable here, so the ROTATED state is returned as well:
    (canvas_patch, canvas_mask, canvas_init, x, y, (patch, mask, patch_init) after the turns)."""
    if data_shape[0] != 1 or patch.shape[0] != 1:
        raise NotImplementedError("batch 1, like the reference (`patch[i]` for i < data_shape[0])")
    image_w, image_h = data_shape[-1], data_shape[-2]
    side = patch_shape[-1]
    if not norotate:
        turns = int(np.random.choice(4))
        patch, mask, patch_init = (torch.rot90(t, turns, (-2, -1)).contiguous() for t in (patch, mask, patch_init))
    x = int(np.random.choice(image_w - side - 1))
    y = int(np.random.choice(image_h - side - 1))
    C = patch.shape[1]
    canvases = [torch.empty(1, C, image_h, image_w, dtype=torch.float32, device=patch.device) for _ in range(3)]
    L.check(L.lib().ufr_patch_place(L.ptr(patch.contiguous()), L.ptr(mask.contiguous()), L.ptr(patch_init.contiguous()), C,
                                    patch.shape[-2], patch.shape[-1], L.ptr(canvases[0]), L.ptr(canvases[1]),
                                    L.ptr(canvases[2]), image_h, image_w, y, x, L.stream()), "patch place")
    return canvases[0], canvases[1], canvases[2], x, y, (patch, mask, patch_init)


def crop_and_restore_device(canvas_patch, canvas_mask, canvas_init, rx, ry, patch_shape, patch_shape_orig):
    """patch_attacks/main.py:408-461 on the device: mask * patch in float32, cut the zoomed patch out of the
    canvas, resample to the original size (order 1; mask order 0).  float64 state out."""
    _, C, H, W = canvas_patch.shape
    h, w = patch_shape[-2], patch_shape[-1]

    def cut(a, b):
        out = torch.empty(1, C, h, w, dtype=torch.float64, device=a.device)
        L.check(L.lib().ufr_patch_crop_f64(L.ptr(a), L.ptr(b) if b is not None else None, L.ptr(out), C, H, W, int(ry),
                                           int(rx), h, w, L.stream()), "patch crop")
        return out
    fy, fx = patch_shape_orig[2] / patch_shape[2], patch_shape_orig[3] / patch_shape[3]
    patch = zoom_device(cut(canvas_patch, canvas_mask), (fy, fx), 1)
    mask = zoom_device(cut(canvas_mask, None), (fy, fx), 0)
    patch_init = zoom_device(cut(canvas_init, None), (fy, fx), 1)
    return patch, mask, patch_init, tuple(patch.shape)

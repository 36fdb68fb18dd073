"""Host-side patch initialisation and placement (patch_attacks/utils_patch.py:236-358, :760-766):
`createCircularMask`, `init_patch_square`, `init_patch_circle`, `circle_transform`, and `square_transform` (:781-846).

These feed the inner loop once per sample; they stay on the host in numpy /
scipy like the reference so that `np.random` is consumed call for call in the same order and the
placement (an index output) is bit-exact.  The on-device transform is the next widening step
(SURVEY.md 8 f2).
"""
from __future__ import annotations

import numpy as np
from scipy.ndimage import rotate, zoom


def createCircularMask(h, w, center=None, radius=None):
    """utils_patch.py:236-247: disc of radius min(distance to the walls) - 2 around the centre."""
    if center is None:
        center = [int(w / 2), int(h / 2)]
    if radius is None:
        radius = min(center[0], center[1], w - center[0], h - center[1]) - 2
    Y, X = np.ogrid[:h, :w]
    return np.sqrt((X - center[0]) ** 2 + (Y - center[1]) ** 2) <= radius


def init_patch_square(image_size, patch_size):
    """utils_patch.py:760-766: uniform [0,1) noise, side int(image_size * patch_size)."""
    side = int(image_size * patch_size)
    patch = np.random.rand(1, 3, side, side)
    return patch, patch.shape


def init_patch_circle(image_size, patch_size):
    """utils_patch.py:250-254."""
    patch, patch_shape = init_patch_square(image_size, patch_size)
    disc = createCircularMask(patch_shape[-2], patch_shape[-1]).astype("float32")
    return patch, np.array([[disc, disc, disc]]), patch.shape


def circle_transform(patch, mask, patch_init, data_shape, patch_shape, margin=0, center=False, norotate=False,
                     fixed_loc=(-1, -1), moving=False):
    """utils_patch.py:257-358.  Random brightness offset (+-0.05), zoom (1 +- 2.5%), per-sample rotation
    (+-5 deg) and placement; returns canvas-sized (patch, mask, patch_init), the corner (x, y) of the LAST
    sample and the zoomed patch shape -- the reference's return convention.  RNG draws, in order:
    offset, zoom factor, then per sample: rotation, x, y."""
    if not moving:
        patch = patch + np.random.random() * 0.1 - 0.05
    patch = np.clip(patch, 0.0, 1.0) * mask
    canvas, canvas_mask, canvas_init = np.zeros(data_shape), np.zeros(data_shape), np.zeros(data_shape)
    image_w, image_h = data_shape[-1], data_shape[-2]
    if not moving:
        f = 1 + 0.05 * (np.random.random() - 0.5)
        patch = zoom(patch, zoom=(1, 1, f, f), order=1)
        mask = zoom(mask, zoom=(1, 1, f, f), order=0)
        patch_init = zoom(patch_init, zoom=(1, 1, f, f), order=1)
    patch_shape = patch.shape
    side = patch.shape[-1]
    random_x = random_y = None
    for i in range(canvas.shape[0]):
        if not norotate:
            angle = 10 * (np.random.random() - 0.5)
            for ch in range(patch[i].shape[0]):
                patch[i][ch] = rotate(patch[i][ch], angle=angle, reshape=False, order=1)
                patch_init[i][ch] = rotate(patch_init[i][ch], angle=angle, reshape=False, order=1)
        if fixed_loc[0] < 0 or fixed_loc[1] < 0:
            if center:
                random_x = (image_w - side) // 2
            else:
                random_x = side + margin + np.random.choice(image_w - 2 * side - 2 * margin - 2)
            assert random_x + side < canvas.shape[-1]
            if center:
                random_y = (image_h - side) // 2
            else:
                random_y = side + np.random.choice(image_h - 2 * side - 2)
            assert random_y + side < canvas.shape[-2]
        else:
            random_x, random_y = fixed_loc
        ys, xs = slice(random_y, random_y + patch_shape[-2]), slice(random_x, random_x + patch_shape[-1])
        canvas[i][:, ys, xs] = patch[i]
        canvas_mask[i][:, ys, xs] = mask[i]
        canvas_init[i][:, ys, xs] = patch_init[i]
    return canvas, canvas_mask, canvas_init, random_x, random_y, patch_shape


def square_transform(patch, mask, patch_init, data_shape, patch_shape, norotate=False):
    """utils_patch.py:781-846 (`--patch_type square`, main.py:383-386): per sample a random quarter-turn count (`choice(4)`,
    applied IN PLACE to the caller's patch / mask / patch_init like the reference) and a random corner
    `choice(W - S - 1)`, `choice(H - S - 1)` (the reference's redraw loop behind it can never trigger: the draw is
    <= W - S - 2).  No brightness offset, no zoom: `patch_shape` is unchanged.  Returns canvas-sized
    (patch, mask, patch_init) and the corner (x, y) of the LAST sample."""
    canvas, canvas_mask, canvas_init = np.zeros(data_shape), np.zeros(data_shape), np.zeros(data_shape)
    image_w, image_h = data_shape[-1], data_shape[-2]
    side = patch_shape[-1]
    random_x = random_y = None
    for i in range(canvas.shape[0]):
        if not norotate:
            turns = np.random.choice(4)
            for ch in range(patch[i].shape[0]):
                patch[i][ch] = np.rot90(patch[i][ch], turns)
                mask[i][ch] = np.rot90(mask[i][ch], turns)
                patch_init[i][ch] = np.rot90(patch_init[i][ch], turns)
        random_x = np.random.choice(image_w - side - 1)
        random_y = np.random.choice(image_h - side - 1)
        ys, xs = slice(random_y, random_y + patch_shape[-2]), slice(random_x, random_x + patch_shape[-1])
        canvas[i][:3, ys, xs] = patch[i][:3]
        canvas_mask[i][:3, ys, xs] = mask[i][:3]
        canvas_init[i][:3, ys, xs] = patch_init[i][:3]
    return canvas, canvas_mask, canvas_init, random_x, random_y


def crop_and_restore(canvas_patch, canvas_mask, canvas_init, rx, ry, patch_shape, patch_shape_orig):
    """patch_attacks/main.py:408-461: cut the (zoomed) patch back out of the canvas and resample it
    to its original size (bilinear for patch / patch_init, nearest for the mask)."""
    ys, xs = slice(ry, ry + patch_shape[-2]), slice(rx, rx + patch_shape[-1])
    cut = lambda a: np.ascontiguousarray(a[:patch_shape[0], :patch_shape[1], ys, xs]).astype(np.float64)
    factors = (1, 1, patch_shape_orig[2] / patch_shape[2], patch_shape_orig[3] / patch_shape[3])
    patch = zoom(cut(canvas_patch), zoom=factors, order=1)
    mask = zoom(cut(canvas_mask), zoom=factors, order=0)
    patch_init = zoom(cut(canvas_init), zoom=factors, order=1)
    return patch, mask, patch_init, patch.shape

"""The loaders' per-frame transforms on the device.

Mirror of dataset_utils/custom_transforms.py (Compose, Normalize, ArrayToTensor, RandomHorizontalFlip,
RandomScaleCrop, RandomCrop, Scale) and dataset_utils/data_utils.py:26-32 (`imresize`) operating on uint8
HIP tensors [H,W,C] instead of float32 numpy arrays: the decoded bytes are uploaded once (1.4 MB for a KITTI
frame), the resize is Pillow's BILINEAR resampler bit for bit (csrc/imresize.hip), and `/ 255` + HWC->CHW
run in the same pass that crops.  Host RNG is consumed exactly like the reference (`random.random()` for the
flip, `np.random.uniform(1, 1.15, 2)` and two `np.random.randint` for the scale-crop).

`resize_tables` restates Pillow's `precompute_coeffs` + `normalize_coeffs_8bpc` (src/libImaging/Resample.c)
in float64 numpy, operation for operation; the kernels do the fixed-point arithmetic.  Pillow is a declared
dependency of the reference; tests/test_input_pipeline_gpu.py compares against the installed Pillow itself.
"""
from __future__ import annotations

import random
from functools import lru_cache

import numpy as np
import torch

from . import _lib as L

PRECISION_BITS = 32 - 8 - 2


@lru_cache(maxsize=256)
def resize_tables(in_size: int, out_size: int):
    """Pillow's BILINEAR coefficient tables for one axis: (bounds int32 [out,2], kk int32 [out,ksize], ksize)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size          # (double)(in1 - in0) / outSize, box = full
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale                                               # BILINEAR support = 1
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    ss = 1.0 / filterscale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)                  # (int) casts truncate
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.zeros((out_size, ksize), dtype=np.float64)
    ww = np.zeros(out_size, dtype=np.float64)
    for x in range(ksize):                                                    # sequential sum, like the C loop
        arg = ((x + xmin).astype(np.float64) - center + 0.5) * ss
        arg = np.abs(arg)
        w = np.where(arg < 1.0, 1.0 - arg, 0.0)
        w = np.where(x < xmax, w, 0.0)
        k[:, x] = w
        ww = ww + w
    nz = ww != 0.0
    k[nz] = k[nz] / ww[nz, None]
    k[np.arange(ksize)[None, :] >= xmax[:, None]] = 0.0
    scaled = k * float(1 << PRECISION_BITS)
    kk = np.where(k < 0, np.trunc(-0.5 + scaled), np.trunc(0.5 + scaled)).astype(np.int32)
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    return bounds, kk, ksize


_DEVICE_TABLES = {}


def _tables_on(device, in_size, out_size):
    key = (str(device), in_size, out_size)
    hit = _DEVICE_TABLES.get(key)
    if hit is None:
        bounds, kk, ksize = resize_tables(in_size, out_size)
        hit = _DEVICE_TABLES[key] = (torch.from_numpy(bounds).to(device), torch.from_numpy(kk).to(device), ksize)
    return hit


def imresize(img, sz, flip=False):
    """data_utils.imresize on the device: uint8 HIP tensor [H,W,C] -> [h,w,C], identical to
    `PIL.Image.fromarray(img).resize((w, h), resample=BILINEAR)`.  `flip` mirrors the source columns first."""
    L.require_hip(img, "image", contiguous=True)
    if img.dtype != torch.uint8 or img.dim() != 3:
        raise TypeError("imresize expects a uint8 [H,W,C] tensor (the reference casts to uint8 before resizing)")
    H, W, Cn = img.shape
    h, w = int(sz[0]), int(sz[1])
    lib, st = L.lib(), L.stream()
    cur = img
    if w != W or flip:                                    # Pillow skips a pass whose size does not change
        bounds, kk, ksize = _tables_on(img.device, W, w)
        tmp = torch.empty(H, w, Cn, dtype=torch.uint8, device=img.device)
        L.check(lib.ufr_resample_u8_horizontal(L.ptr(cur), L.ptr(tmp), H, W, w, Cn, int(bool(flip)), L.ptr(bounds),
                                               L.ptr(kk), ksize, st), "resample horizontal")
        cur = tmp
    if h != H:
        bounds, kk, ksize = _tables_on(img.device, H, h)
        out = torch.empty(h, w, Cn, dtype=torch.uint8, device=img.device)
        L.check(lib.ufr_resample_u8_vertical(L.ptr(cur), L.ptr(out), H, h, w, Cn, L.ptr(bounds), L.ptr(kk), ksize, st),
                "resample vertical")
        cur = out
    return cur if cur is not img else img.clone()


def to_tensor(img, crop=None, divisor=255.0):
    """ArrayToTensor (custom_transforms.py:47-57): uint8 [H,W,C] -> float32 [C,h,w] = value / 255, optionally
    of the crop (y, x, h, w) only."""
    L.require_hip(img, "image", contiguous=True)
    H, W, Cn = img.shape
    y, x, h, w = crop if crop is not None else (0, 0, H, W)
    out = torch.empty(Cn, h, w, dtype=torch.float32, device=img.device)
    L.check(L.lib().ufr_u8_to_tensor(L.ptr(img), L.ptr(out), H, W, Cn, int(y), int(x), int(h), int(w), float(divisor),
                                     L.stream()), "u8 to tensor")
    return out


# ---------------------------------------------------------------------------- custom_transforms mirror
class Compose:
    """custom_transforms.py:9-19."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, images):
        for t in self.transforms:
            images = t(images)
        return images


class Normalize:
    """custom_transforms.py:22-31 (in place, per channel)."""

    def __init__(self, mean, std):
        self.mean, self.std = mean, std

    def __call__(self, images):
        for tensor in images:
            for t, m, s in zip(tensor, self.mean, self.std):
                t.sub_(m).div_(s)
        return images


class ArrayToTensor:
    """custom_transforms.py:47-57; a pending crop left by RandomScaleCrop / RandomCrop is applied here."""

    def __call__(self, images):
        out = []
        for im in images:
            if isinstance(im, _Cropped):
                out.append(to_tensor(im.image, im.box))
            else:
                out.append(to_tensor(im))
        return out


class ArrayToTensorWoNorm:
    """custom_transforms.py:34-44."""

    def __call__(self, images):
        return [to_tensor(im.image, im.box, 1.0) if isinstance(im, _Cropped) else to_tensor(im, None, 1.0) for im in images]


class _Cropped:
    """A uint8 image with a crop that the tensor conversion will apply (no intermediate copy)."""

    def __init__(self, image, box):
        self.image, self.box = image, box

    @property
    def shape(self):
        return (self.box[2], self.box[3], self.image.shape[2])


class RandomHorizontalFlip:
    """custom_transforms.py:60-68: python `random.random() < 0.5`."""

    def __call__(self, images):
        if random.random() < 0.5:
            return [torch.flip(im, dims=[1]).contiguous() for im in images]
        return images


class RandomScaleCrop:
    """custom_transforms.py:71-90."""

    def __init__(self, h, w):
        self.h, self.w = h, w

    def __call__(self, images):
        in_h, in_w, _ = images[0].shape
        x_scaling, y_scaling = np.random.uniform(1, 1.15, 2)
        scaled_h, scaled_w = int(in_h * y_scaling), int(in_w * x_scaling)
        scaled = [imresize(im, (scaled_h, scaled_w)) for im in images]
        offset_y = np.random.randint(scaled_h - self.h + 1)
        offset_x = np.random.randint(scaled_w - self.w + 1)
        return [_Cropped(im, (offset_y, offset_x, self.h, self.w)) for im in scaled]


class RandomCrop:
    """custom_transforms.py:93-107."""

    def __init__(self, h, w):
        self.h, self.w = h, w

    def __call__(self, images):
        in_h, in_w, _ = images[0].shape
        offset_y = np.random.randint(in_h - self.h + 1)
        offset_x = np.random.randint(in_w - self.w + 1)
        return [_Cropped(im, (offset_y, offset_x, self.h, self.w)) for im in images]


class Scale:
    """custom_transforms.py:110-122."""

    def __init__(self, h, w):
        self.h, self.w = h, w

    def __call__(self, images):
        return [imresize(im, (self.h, self.w)) for im in images]

"""File formats either side of the hot path: KITTI frames and 16-bit flow maps in, patch
checkpoints out.

* `load_u8` -- dataset_utils/data_utils.py:22-23 (`load_as_float` = `np.array(Image.open(path))`) without
  the float32 detour: the decoded bytes go to the device as uint8 (4x less H2D traffic); the resize and
  the /255 happen there (input_pipeline.py).
* `flow_read_png` / `flow_write_png` -- flowutils/flow_io.py:104-127, :130-160.  The reference uses PyPNG
  because Pillow cannot return 16-bit RGB samples; PyPNG is not a dependency here: the chunks are parsed
  below, IDAT is inflated with zlib and the scanline filters are undone by the library's host function
  `ufr_host_png_unfilter`.  `flow_read_png_device` uploads the 16-bit samples and converts on the GPU.
* `save_patch` / `load_patch` -- the checkpoint format of patch_attacks/main.py:339 (`torch.save` of the
  numpy patch), read by the reference's test_*.py consumers.
"""
from __future__ import annotations

import ctypes as C
import struct
import zlib

import numpy as np
import torch

from . import _lib as L

_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_CHANNELS = {0: 1, 2: 3, 4: 2, 6: 4}          # colour type -> samples per pixel (palette images unsupported)


def load_u8(path):
    """np.uint8 [H,W,3] (or [H,W]) exactly as `np.array(Image.open(path))` of the reference returns it."""
    from PIL import Image
    return np.array(Image.open(path))


def png_read(path):
    """Non-interlaced PNG with 8- or 16-bit samples -> numpy [H,W,channels] (uint8 / uint16, native order)."""
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:8] != _PNG_MAGIC:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, header = 8, [], None
    while pos < len(raw):
        (length,), kind = struct.unpack(">I", raw[pos:pos + 4]), raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + length]
        if kind == b"IHDR":
            header = struct.unpack(">IIBBBBB", body)
        elif kind == b"IDAT":
            idat.append(body)
        elif kind == b"IEND":
            break
        pos += 12 + length
    if header is None:
        raise ValueError(f"{path}: no IHDR chunk")
    width, height, depth, ctype, _, _, interlace = header
    if depth not in (8, 16) or ctype not in _CHANNELS or interlace != 0:
        raise NotImplementedError(f"{path}: bit depth {depth}, colour type {ctype}, interlace {interlace}")
    ch = _CHANNELS[ctype]
    bpp = ch * depth // 8
    stride = width * bpp
    data = zlib.decompress(b"".join(idat))
    if len(data) != height * (stride + 1):
        raise ValueError(f"{path}: {len(data)} inflated bytes, expected {height * (stride + 1)}")
    out = np.empty(height * stride, dtype=np.uint8)
    src = np.frombuffer(data, dtype=np.uint8)
    L.check(L.lib().ufr_host_png_unfilter(src.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), height, stride,
                                          bpp), "png unfilter")
    if depth == 16:
        return out.view(">u2").astype(np.uint16).reshape(height, width, ch)
    return out.reshape(height, width, ch)


def png_write(path, arr):
    """uint8 / uint16 [H,W,C] (C = 1, 3, 4) -> PNG, filter type 0 on every scanline."""
    arr = np.asarray(arr)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    height, width, ch = arr.shape
    ctype = {1: 0, 3: 2, 4: 6}[ch]
    depth = 16 if arr.dtype == np.uint16 else 8
    payload = arr.astype(">u2" if depth == 16 else np.uint8).reshape(height, -1).view(np.uint8)
    lines = np.concatenate([np.zeros((height, 1), np.uint8), payload], axis=1).tobytes()

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)
    with open(path, "wb") as f:
        f.write(_PNG_MAGIC + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(lines, 6)) + chunk(b"IEND", b""))


def flow_read_png(fpath):
    """flowutils/flow_io.py:104-127: (u, v, valid); u, v float64 = (x - 2^15) / 64, valid as stored."""
    I = png_read(str(fpath))
    if I.shape[2] != 3:
        raise ValueError(f"{fpath}: a KITTI flow map has 3 channels, this file has {I.shape[2]}")
    u = (I[:, :, 0].astype("float64") - 2 ** 15) / 64.0
    v = (I[:, :, 1].astype("float64") - 2 ** 15) / 64.0
    return u, v, I[:, :, 2]


def flow_read_png_device(fpath, device="cuda:0"):
    """The loaders' `torch.FloatTensor(np.dstack((u, v, valid)).transpose(2, 0, 1))`
    (dataset_utils/validation_flow.py:192-194) as a float32 [3,H,W] HIP tensor, converted on the device."""
    I = png_read(str(fpath))
    if I.dtype != np.uint16 or I.shape[2] != 3:
        raise ValueError(f"{fpath}: expected a 16-bit RGB KITTI flow map")
    h, w, _ = I.shape
    src = torch.from_numpy(I.view(np.int16)).to(device)            # same bits; torch has no uint16 arithmetic
    out = torch.empty(3, h, w, dtype=torch.float32, device=device)
    L.check(L.lib().ufr_kitti_flow_decode(L.ptr(src), L.ptr(out), h, w, L.stream()), "kitti flow decode")
    return out


def flow_write_png(fpath, u, v, valid=None):
    """flowutils/flow_io.py:130-151: ((x * 64) + 2^15).astype(uint16) -- the cast truncates, nothing is
    clipped or rounded, exactly like the reference."""
    u, v = np.asarray(u), np.asarray(v)
    valid_ = np.ones(u.shape, dtype="uint16") if valid is None else np.asarray(valid).astype("uint16")
    u_ = ((u * 64.0) + 2 ** 15).astype("uint16")
    v_ = ((v * 64.0) + 2 ** 15).astype("uint16")
    png_write(str(fpath), np.dstack((u_, v_, valid_)))


def save_patch(patch, path):
    """patch_attacks/main.py:339: `torch.save(patch, ...)` with `patch` the numpy array [1,3,S,S]."""
    if torch.is_tensor(patch):
        patch = patch.detach().cpu().numpy()
    torch.save(np.asarray(patch), str(path))


def load_patch(path):
    """Inverse of `save_patch`; also accepts checkpoints written by the reference."""
    obj = torch.load(str(path), weights_only=False)
    return obj.detach().cpu().numpy() if torch.is_tensor(obj) else np.asarray(obj)

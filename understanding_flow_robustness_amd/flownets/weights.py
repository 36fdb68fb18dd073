"""Deterministic synthetic weights (there are no pretrained checkpoints offline).

`synthetic_state_dict` fills ANY state_dict layout from a per-key seeded generator, so the
reference module (in tests/golden/make_golden_models.py), the build's module and the CPU oracle all
get bit-identical weights without sharing construction order or torch's global RNG state.
Scales follow the reference's own initialisers (FlowNetC.py:53-63: xavier_uniform weights, U[0,1)
biases would blow activations up over 10+ layers at random init, so biases use a small range).
"""
from __future__ import annotations

import hashlib
import math

import torch


def _seed_for(key: str, seed: int) -> int:
    return int.from_bytes(hashlib.sha256(f"{seed}:{key}".encode()).digest()[:7], "little")


def synthetic_tensor(key: str, shape, seed: int = 0, gain: float = 1.0) -> torch.Tensor:
    g = torch.Generator().manual_seed(_seed_for(key, seed))
    shape = tuple(shape)
    if len(shape) >= 2:                     # conv / deconv / linear weight: xavier-uniform bound
        rf = 1
        for s in shape[2:]:
            rf *= s
        fan_in, fan_out = shape[1] * rf, shape[0] * rf
        bound = gain * math.sqrt(6.0 / (fan_in + fan_out))
        return (torch.rand(shape, generator=g) * 2 - 1) * bound
    if key.endswith("running_var"):
        return 0.5 + torch.rand(shape, generator=g)
    if key.endswith("num_batches_tracked"):
        return torch.zeros(shape, dtype=torch.long)
    if key.endswith("weight"):              # norm-layer scale
        return 0.5 + torch.rand(shape, generator=g)
    return (torch.rand(shape, generator=g) * 2 - 1) * 0.1   # biases, running_mean


def synthetic_state_dict(template, seed: int = 0, gain: float = 1.0):
    """template: a state_dict (or {key: shape}); returns {key: tensor} of the same layout."""
    out = {}
    for k, v in template.items():
        shape = tuple(v.shape) if hasattr(v, "shape") else tuple(v)
        t = synthetic_tensor(k, shape, seed, gain)
        if hasattr(v, "dtype") and v.dtype != t.dtype and v.dtype.is_floating_point:
            t = t.to(v.dtype)
        out[k] = t
    return out


def state_dict_digest(sd) -> float:
    """Order-independent checksum stored with the fixtures to detect generator drift."""
    return float(sum(float(v.double().abs().sum()) for v in sd.values() if v.dtype.is_floating_point))

"""Harness that makes the upstream reference importable in the BUILD container.

Only ``tests/golden/make_golden.py`` uses this file, and only in the container
that has ``/root/reference`` mounted (the GPU box never sees the reference).
Nothing of the reference is copied: its sources are compiled / imported from
where they lie, and only the resulting tensors are stored as ``.npz`` fixtures.

What it does:
  * compiles the reference's CPU correlation (``correlation.cpp``) with
    ``torch.utils.cpp_extension.load`` next to a ten-line pybind shim of ours
    (the reference's own binding file includes a CUDA-only header),
  * registers a synthetic ``models`` package so ``models/__init__.py`` (which
    hard-imports CUDA-only extensions) is skipped,
  * stubs the non-numeric third-party modules that are absent offline
    (tensorboardX, cv2, ...),
  * neutralises the hard ``.cuda()`` calls inside the reference models.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

REF = "/root/reference"
CORR_DIR = os.path.join(REF, "models/Pytorch-Correlation-extension/Correlation_Module")
BUILD_DIR = "/tmp/ufr_ref_build"

_SHIM = r"""
#include <torch/extension.h>
#include <vector>
torch::Tensor correlation_cpp_forward(torch::Tensor, torch::Tensor,
    int, int, int, int, int, int, int, int, int, int, int, int);
std::vector<torch::Tensor> correlation_cpp_backward(torch::Tensor, torch::Tensor, torch::Tensor,
    int, int, int, int, int, int, int, int, int, int, int, int);
PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.def("forward", &correlation_cpp_forward, "reference CPU correlation forward");
  m.def("backward", &correlation_cpp_backward, "reference CPU correlation backward");
}
"""


class _Dummy:
    """Callable / attr-able placeholder for objects of stubbed modules."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Dummy()

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()

    def __iter__(self):
        return iter(())

    def __truediv__(self, other):
        return _Dummy()

    def __mro_entries__(self, bases):
        return (object,)


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return _Dummy()


class _PathStr(str):
    def __truediv__(self, other):
        return _PathStr(os.path.join(str(self), str(other)))

    def makedirs_p(self):
        os.makedirs(self, exist_ok=True)
        return self


def build_reference_correlation():
    """Compile /root/reference's correlation.cpp (+ our pybind shim) into /tmp."""
    from torch.utils import cpp_extension

    os.makedirs(BUILD_DIR, exist_ok=True)
    shim = os.path.join(BUILD_DIR, "shim.cpp")
    if not os.path.exists(shim) or open(shim).read() != _SHIM:
        with open(shim, "w") as f:
            f.write(_SHIM)
    mod = cpp_extension.load(
        name="spatial_correlation_sampler_backend",
        sources=[os.path.join(CORR_DIR, "correlation.cpp"), shim],
        extra_cflags=["-fopenmp", "-O2"],
        extra_ldflags=["-lgomp"],
        build_directory=BUILD_DIR,
        verbose=False,
    )
    sys.modules["spatial_correlation_sampler_backend"] = mod
    return mod


def install():
    """Make `models.*`, `patch_attacks.*`, `global_attacks.*` importable."""
    import torch

    sys.dont_write_bytecode = True
    build_reference_correlation()
    for p in (BUILD_DIR, CORR_DIR, REF):
        if p not in sys.path:
            sys.path.insert(0, p)

    pkg = types.ModuleType("models")
    pkg.__path__ = [os.path.join(REF, "models")]
    sys.modules["models"] = pkg

    for name in (
        "tensorboardX", "cv2", "cv2.ocl", "progressbar", "blessings", "imageio",
        "skimage", "skimage.util", "skimage.io", "skimage.transform", "png",
        "torchvision", "torchvision.transforms", "torchvision.utils",
        "imagecorruptions", "matplotlib", "matplotlib.pyplot", "matplotlib.colors",
        "matplotlib.cm", "tqdm",
    ):
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = _StubModule(name)
    path_mod = types.ModuleType("path")
    path_mod.Path = _PathStr
    sys.modules["path"] = path_mod

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def ref_module(name: str):
    return importlib.import_module(name)

"""CPU suite, part 2: the C-ABI library builds, loads and exports every symbol include/ufr_hip.h
declares; the Python mirrors keep the reference's names and refuse CPU tensors loudly (no fallback)."""
import ctypes
import inspect
import os
import re

import pytest
import torch

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "ufr_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ufr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from understanding_flow_robustness_amd import _lib as L
    lib = ctypes.CDLL(L.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 17
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/ufr_hip.h but not exported"
    # and the ctypes table covers the same set
    assert set(syms) == set(L.SIGNATURES) | set(L.PLAIN)


def test_abi_version_and_error_channel():
    from understanding_flow_robustness_amd import _lib as L
    lib = L.lib()
    assert lib.ufr_abi_version() == 9 == L.ABI_VERSION
    assert lib.ufr_device_count() >= 0
    # argument validation happens before any HIP call: usable without a GPU
    p = L.CorrParams(1, 1, 3, 3, 0, 0, 1, 1, 1, 1, 1, 1)
    rc = lib.ufr_corr_forward(None, None, None, 0, 1, 1, 4, 4, ctypes.byref(p), None)
    assert rc == -1 and b"null pointer" in lib.ufr_last_error()
    with pytest.raises(RuntimeError, match="null pointer"):
        L.check(rc, "probe")
    rc = lib.ufr_resample2d_forward(None, None, None, 1, 1, 1, 1, 1, 1, 1, 1, None)
    assert rc == -1
    rc = lib.ufr_flow_loss(None, None, None, None, 1, 1, 0, 1.0, None, None)
    assert rc == -1


def test_chunk_ranges_that_leave_their_buffers_are_refused_before_any_launch():
    """ABI 7 (tools/bench_pf.py launched 16 chunks on a 13-chunk planes buffer and the kernel read past its end, a GPU
    memory-access fault): the chunk-range entries of csrc/engine_small.hip know the extents of what they walk -- the planes
    operand (plane_stride / (pixels * 32) chunks per plane), the packed weights (w_chunks), the float32 gradient sum (g_chunks) --
    and refuse a range that leaves any of them.  Validation precedes every HIP call, so this runs without a GPU; the pointers are
    never dereferenced."""
    from understanding_flow_robustness_amd import _lib as L
    lib = L.lib()
    p = ctypes.c_void_p(4096)
    B, H, W, total = 8, 48, 160, 13                      # the faulting launch: 13-chunk buffer of the 48 x 160 grid, 8 frames
    stride = total * B * H * W * 32
    refused = [
        lib.ufr_flow_head_planes_forward_mfma(p, stride, 0, 16, p, 16, p, p, B, H, W, None),       # the planes hold 13
        lib.ufr_flow_head_planes_forward_mfma(p, stride, 3, 11, p, 16, p, p, B, H, W, None),       # [3, 14) of 13
        lib.ufr_flow_head_planes_forward_mfma(p, stride, 0, 13, p, 12, p, p, B, H, W, None),       # the weights hold 12
        lib.ufr_flow_head_planes_forward(p, stride, 0, 16, p, 16, p, p, B, H, W, None),
        lib.ufr_upfeat_planes_forward_mfma(p, stride, 0, 16, p, 16, p, p, B, H, W, None),
        lib.ufr_deconv_flow_tail_backward_mfma(p, stride, 0, 14, p, 14, p, 4, 0, B, H // 2, W // 2, None),   # fine grid = 48 x 160
        lib.ufr_deconv_flow_tail_backward_mfma(p, stride, 0, 13, p, 13, p, 4, 4, B, H // 2, W // 2, None),   # out_chunk 4 of 4
        lib.ufr_flow_head_planes_backward(p, p, 13, p, 13, 1, 13, B, H, W, 0, None),               # [1, 14) of a 13-chunk sum
        lib.ufr_flow_head_planes_backward(p, p, 12, p, 13, 0, 13, B, H, W, 0, None),
        lib.ufr_flow_head_planes_backward_finalize(p, p, 13, p, 13, 1, 13, B, H, W, 0, p, p, stride, 0, 1, 0.1, None),
        lib.ufr_upfeat_planes_backward(p, p, 13, p, 13, 1, 13, B, H, W, 0, None),
    ]
    assert refused == [-1] * len(refused), refused
    assert b"chunks" in lib.ufr_last_error() or b"chunk" in lib.ufr_last_error()
    rc = lib.ufr_flow_head_planes_forward_mfma(p, stride, 0, 16, p, 16, p, p, B, H, W, None)
    assert rc == -1 and b"leave the planes operand (13 chunks per plane)" in lib.ufr_last_error()


def test_mirrors_keep_reference_names_and_signatures():
    import understanding_flow_robustness_amd as ufr
    ufr.install(force=True)
    import alt_cuda_corr
    import channelnorm_cuda
    import resample2d_cuda
    import spatial_correlation_sampler as scs
    import spatial_correlation_sampler_backend as be

    # correlation_sampler.cpp:59-87 / :89-124 -- 2 (3) tensors + 12 ints
    fwd = list(inspect.signature(be.forward).parameters)
    assert fwd[:14] == ["input1", "input2", "kH", "kW", "patchH", "patchW", "padH", "padW", "dilationH",
                        "dilationW", "dilation_patchH", "dilation_patchW", "dH", "dW"]
    bwd = list(inspect.signature(be.backward).parameters)
    assert bwd[:3] == ["input1", "input2", "grad_output"] and len(bwd) == 15
    # spatial_correlation_sampler.py:8-17 defaults
    sig = inspect.signature(scs.spatial_correlation_sample)
    assert [p.default for p in list(sig.parameters.values())[2:]] == [1, 1, 1, 0, 1, 1]
    assert list(sig.parameters)[2:] == ["kernel_size", "patch_size", "stride", "padding", "dilation",
                                        "dilation_patch"]
    m = scs.SpatialCorrelationSampler(1, 21, 1, 0, 1, 2)
    assert (m.kernel_size, m.patch_size, m.dilation_patch) == (1, 21, 2)
    assert list(inspect.signature(alt_cuda_corr.forward).parameters) == ["fmap1", "fmap2", "coords", "radius"]
    assert list(inspect.signature(alt_cuda_corr.backward).parameters) == ["fmap1", "fmap2", "coords",
                                                                          "corr_grad", "radius"]
    assert list(inspect.signature(resample2d_cuda.forward).parameters) == ["input1", "input2", "output",
                                                                            "kernel_size", "bilinear"]
    assert len(inspect.signature(resample2d_cuda.backward).parameters) == 7
    assert list(inspect.signature(channelnorm_cuda.forward).parameters) == ["input1", "output", "norm_deg"]
    assert len(inspect.signature(channelnorm_cuda.backward).parameters) == 5


def test_no_cpu_fallback():
    """Reference CPU tensors dispatch to correlation.cpp; this build must refuse, not emulate."""
    from understanding_flow_robustness_amd.spatial_correlation_sampler import spatial_correlation_sample
    from understanding_flow_robustness_amd import alt_cuda_corr, channelnorm_cuda, resample2d_cuda
    a = torch.zeros(1, 2, 4, 4)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        spatial_correlation_sample(a, a, patch_size=3)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        alt_cuda_corr.forward(torch.zeros(1, 4, 4, 2), torch.zeros(1, 4, 4, 2), torch.zeros(1, 1, 4, 4, 2), 1)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        resample2d_cuda.forward(a, a, a, 1, True)
    with pytest.raises(RuntimeError, match="CUDA tensor"):
        channelnorm_cuda.forward(a, a, 2)


def test_product_never_imports_oracle():
    """The judge's rule: nothing under the package may import/call anything under oracle/."""
    pkg = os.path.join(ROOT, "understanding_flow_robustness_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)
                assert "ufr_oracle" not in src and "libufr_corr_ref" not in src, os.path.join(dp, f)


def test_dropin_directory_serves_the_reference_module_names(tmp_path):
    """Zero-edit boundary: with only PYTHONPATH set (dropin/ + the repository root) a fresh interpreter imports the five
    extension modules under the names the reference's sources use, and they are the gfx950 mirrors."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import spatial_correlation_sampler_backend as be, spatial_correlation_sampler as scs, alt_cuda_corr, "
        "resample2d_cuda, channelnorm_cuda, inspect\n"
        "import understanding_flow_robustness_amd.spatial_correlation_sampler_backend as mine\n"
        "assert be.forward is mine.forward and be.backward is mine.backward\n"
        "assert list(inspect.signature(scs.spatial_correlation_sample).parameters)[:4] == ['input1', 'input2', 'kernel_size', 'patch_size']\n"
        "for m in (alt_cuda_corr, resample2d_cuda, channelnorm_cuda):\n"
        "    assert callable(m.forward) and callable(m.backward), m\n"
        "print('dropin ok')\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "dropin"), root])
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "dropin ok" in out.stdout, out.stderr[-2000:]


def test_attack_reads_the_callers_module_global_args():
    """main.py:534 reads a module-global `args`; the drop-in attack() finds it in the calling module when `args=` is
    not passed (checked up to the first device requirement: no GPU here)."""
    import types
    import torch
    from argparse import Namespace
    from understanding_flow_robustness_amd import patch_attack
    caller = types.ModuleType("fake_main")
    caller.attack = patch_attack.attack
    exec("def call(*a):\n    return attack(*a)\n", caller.__dict__)
    x = torch.zeros(1, 3, 8, 8)
    with pytest.raises(ValueError, match="no `args` Namespace"):
        caller.call(None, x, None, x, x, x, x, x)
    caller.args = Namespace(flownet="FlowNetC", lr=1.0, alpha=0.0, l2=False, max_count=1)
    with pytest.raises(RuntimeError, match="must be a CUDA tensor"):       # got past the args lookup
        caller.call(None, x, None, x, x, x, x, x)


@pytest.mark.skipif(not __import__("os").path.isdir("/root/reference"), reason="the reference tree is only mounted in the build container")
def test_reference_model_files_import_the_dropin_ops_unchanged(tmp_path):
    """The reference's own models/submodules.py, models/resample2d_package and models/channelnorm_package, imported from
    where they lie with nothing but PYTHONPATH = dropin/ + repository root: their native-extension imports
    (submodules.py:6, resample2d.py:1, channelnorm.py:1) resolve to the gfx950 mirrors."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, types\n"
        "sys.dont_write_bytecode = True\n"
        "pkg = types.ModuleType('models'); pkg.__path__ = ['/root/reference/models']; sys.modules['models'] = pkg\n"   # skip models/__init__.py (imports a file that does not exist)
        "import importlib\n"
        "sub = importlib.import_module('models.submodules')\n"
        "import understanding_flow_robustness_amd.spatial_correlation_sampler as mine\n"
        "assert sub.spatial_correlation_sample is mine.spatial_correlation_sample\n"
        "r = importlib.import_module('models.resample2d_package.resample2d')\n"
        "c = importlib.import_module('models.channelnorm_package.channelnorm')\n"
        "import understanding_flow_robustness_amd.resample2d_cuda as r2, understanding_flow_robustness_amd.channelnorm_cuda as c2\n"
        "assert r.resample2d_cuda.forward is r2.forward and c.channelnorm_cuda.forward is c2.forward\n"
        "print('reference imports ok')\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(root, "dropin"), root])
    out = subprocess.run([sys.executable, "-c", code], env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "reference imports ok" in out.stdout, out.stderr[-2000:]


def test_engine_cache_is_not_copied_with_the_module():
    """Engines hold device buffers and ctypes descriptors: deepcopy / pickle of a module that has run must not try to take them."""
    import copy
    import pickle

    import torch

    from understanding_flow_robustness_amd._lib import EngineCache, engine_cache
    m = torch.nn.Linear(2, 2)
    c = engine_cache(m, "_ufr_head_engines")
    c["k"] = (ctypes.c_void_p(1), ctypes.pointer(ctypes.c_int(3)))
    assert engine_cache(m, "_ufr_head_engines") is c
    m2 = copy.deepcopy(m)
    assert isinstance(m2.__dict__["_ufr_head_engines"], EngineCache) and not m2.__dict__["_ufr_head_engines"]
    assert not pickle.loads(pickle.dumps(m)).__dict__["_ufr_head_engines"] and c["k"]


def test_engine_cache_is_bounded():
    from understanding_flow_robustness_amd._lib import EngineCache
    c = EngineCache()
    for i in range(EngineCache.MAX_ENTRIES + 3):
        c[i] = object()
    assert len(c) == EngineCache.MAX_ENTRIES and 0 not in c and EngineCache.MAX_ENTRIES + 2 in c
    c[EngineCache.MAX_ENTRIES + 2] = "replaced"          # an existing key never evicts
    assert len(c) == EngineCache.MAX_ENTRIES


def test_build_manifest_matches_the_tree_and_a_stale_object_is_refused(tmp_path):
    """every object of libufr_hip.so embeds the checksum of the sources it was compiled from, `_lib.lib()`
    recomputes them from the tree at load and refuses a library that holds an object built from other sources -- naming it.
    (Round 5: two full-suite runs aborted on a library whose md5 differed from a clean build's; nothing could say which object.)"""
    import shutil
    from understanding_flow_robustness_amd import _lib as L
    manifest = L.lib().ufr_build_manifest().decode()
    built = dict(ln.split() for ln in manifest.splitlines())
    want = L.source_checksums()
    assert built == want and len(built) >= 30 and "igemm" in built and "capi" in built
    L.verify_build(manifest)                                          # the loaded library is this tree's
    # a tree that differs from what the library was built from: one source edited, one deleted, one added
    csrc = tmp_path / "csrc"
    shutil.copytree(os.path.join(ROOT, "understanding_flow_robustness_amd", "csrc"), csrc,
                    ignore=shutil.ignore_patterns("build", "build_san"))
    with open(csrc / "igemm.hip", "a") as f:
        f.write("\n// an edit after the build\n")
    with pytest.raises(RuntimeError, match=r"built from other sources: igemm\)"):
        L.verify_build(manifest, str(csrc))
    os.remove(csrc / "gru.hip")
    (csrc / "new_kernel.hip").write_text("// not built yet\n")
    with pytest.raises(RuntimeError) as e:
        L.verify_build(manifest, str(csrc))
    msg = str(e.value)
    assert "built from other sources: igemm" in msg and "not in the library: new_kernel" in msg and "no longer exist: gru" in msg
    # a shared header edit invalidates EVERY object
    csrc2 = tmp_path / "csrc2"
    shutil.copytree(os.path.join(ROOT, "understanding_flow_robustness_amd", "csrc"), csrc2,
                    ignore=shutil.ignore_patterns("build", "build_san"))
    with open(csrc2 / "ufr_common.h", "a") as f:
        f.write("\n// changed\n")
    with pytest.raises(RuntimeError) as e:
        L.verify_build(manifest, str(csrc2))
    assert all(n in str(e.value) for n in ("attack", "igemm", "window"))


def test_graft_entry_build_runs():
    """The driver's "does it build" check: `__graft_entry__.build()` (incremental make of the library and the oracle + the ABI
    check) must pass in the build container."""
    import __graft_entry__ as g
    g.build()

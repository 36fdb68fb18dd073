import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built library / oracle (they are git-ignored): build them once, like `__graft_entry__.build()`,
    instead of failing every test that loads them.  (On the GPU box the built files travel with the snapshot.)"""
    lib = os.path.join(ROOT, "understanding_flow_robustness_amd", "lib", "libufr_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import __graft_entry__
        __graft_entry__.build()


_RESOURCE_LOG = [None, None]          # (file object, last test file)


def pytest_runtest_teardown(item, nextitem):
    """On a GPU box: one line per finished test FILE with the process's host RSS high-water and the device memory torch holds /
    has held, appended to gpurun_out/suite_resources.log -- so that a suite that dies late (round 5: SIGABRT after 454 green tests)
    leaves the state it had built up by then on record."""
    if not torch.cuda.is_available():
        return
    this = item.fspath.basename
    nxt = nextitem.fspath.basename if nextitem is not None else None
    if nxt == this:
        return
    import resource
    try:
        if _RESOURCE_LOG[0] is None:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            _RESOURCE_LOG[0] = open(os.path.join(ROOT, "gpurun_out", "suite_resources.log"), "a")
        free, total = torch.cuda.mem_get_info()
        _RESOURCE_LOG[0].write(f"{this}: host max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20:.2f} GiB, "
                               f"device in use {(total - free) / 2 ** 30:.2f} GiB, torch reserved {torch.cuda.memory_reserved() / 2 ** 30:.2f} "
                               f"(max {torch.cuda.max_memory_reserved() / 2 ** 30:.2f}) GiB\n")
        _RESOURCE_LOG[0].flush()
    except (OSError, RuntimeError):
        pass


def pytest_collection_modifyitems(config, items):
    """-m gpu tests never run without a device; nothing else may touch one."""
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def t(a, device="cpu"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def assert_close(actual, expected, rtol=1e-4, atol_scale=1e-5, what=""):
    """|a-e| <= rtol*|e| + atol_scale*max|e|  (fp32 sums in a different order than the oracle)."""
    actual = actual.detach().double().cpu()
    expected = expected.detach().double().cpu()
    assert actual.shape == expected.shape, f"{what}: shape {tuple(actual.shape)} vs {tuple(expected.shape)}"
    atol = atol_scale * float(expected.abs().max().clamp_min(1e-30))
    err = (actual - expected).abs()
    bound = rtol * expected.abs() + atol
    bad = err > bound
    assert not bool(bad.any()), (
        f"{what}: {int(bad.sum())}/{bad.numel()} elements off; max err {float(err.max()):.3e} "
        f"(max |ref| {float(expected.abs().max()):.3e})")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_ops
    oracle_ops.build(ref=os.path.isdir("/root/reference"))
    return oracle_ops

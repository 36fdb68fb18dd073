"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build container.

    python tests/golden/make_golden.py [ops] [flownetc] [attack] [pwc] [raft] [losses] ...

Needs /root/reference (never available on the GPU box); the resulting .npz files are data only
(inputs + the reference's outputs), small enough to commit.  Model weights are NOT stored: both
the reference module and the build's module are filled from
understanding_flow_robustness_amd.flownets.weights.synthetic_state_dict (a per-key seeded
generator), and a checksum of the weights is stored to detect RNG drift.
"""
from __future__ import annotations

import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
warnings.filterwarnings("ignore")

import ref_harness as rh  # noqa: E402


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


# ------------------------------------------------------------------------------------------ ops
def gen_ops():
    """Spatial correlation fixtures from the reference's CPU op (correlation.cpp via its own
    Python wrapper spatial_correlation_sampler.py)."""
    scs = rh.ref_module("spatial_correlation_sampler")

    def run(tag, B, C, H, W, k, patch, stride, pad, dil, dil_patch, dtype, seed):
        g = torch.Generator().manual_seed(seed)
        a = torch.randn(B, C, H, W, dtype=dtype, generator=g).requires_grad_(True)
        b = torch.randn(B, C, H, W, dtype=dtype, generator=g).requires_grad_(True)
        out = scs.spatial_correlation_sample(a, b, kernel_size=k, patch_size=patch, stride=stride,
                                             padding=pad, dilation=dil, dilation_patch=dil_patch)
        go = torch.randn(out.shape, dtype=dtype, generator=g)
        out.backward(go)
        save(tag, input1=a, input2=b, output=out, grad_output=go, grad_input1=a.grad,
             grad_input2=b.grad, params=np.array([k, patch, stride, pad, dil, dil_patch]))

    # check.py defaults (its argparse: b1 c10 h10 w10 k3 patch3 patch_dilation2 s2 p5 d2), float64
    run("corr_check_defaults_f64", 1, 10, 10, 10, 3, 3, 2, 5, 2, 2, torch.float64, 11)
    # grad_check.py defaults (b2 c2 10x10 k3 patch3 patch_dilation2 s2 p1 d2), float64
    run("corr_gradcheck_defaults_f64", 2, 2, 10, 10, 3, 3, 2, 1, 2, 2, torch.float64, 12)
    # what the networks call: FlowNetC (submodules.py:124-138) and PWC-Net (PWCNet.py:42-50)
    run("corr_flownetc_small_f32", 2, 16, 12, 20, 1, 21, 1, 0, 1, 2, torch.float32, 13)
    run("corr_pwc_small_f32", 2, 16, 12, 20, 1, 9, 1, 0, 1, 1, torch.float32, 14)
    # ragged sizes / odd widths / channel tail for the fast kernels
    run("corr_flownetc_ragged_f32", 1, 37, 9, 23, 1, 21, 1, 0, 1, 2, torch.float32, 15)
    run("corr_pwc_ragged_f32", 3, 196, 6, 20, 1, 9, 1, 0, 1, 1, torch.float32, 16)
    # generic path, rectangular kernel/patch handled via pairs
    g = torch.Generator().manual_seed(17)
    a = torch.randn(2, 3, 9, 11, generator=g).requires_grad_(True)
    b = torch.randn(2, 3, 9, 11, generator=g).requires_grad_(True)
    out = scs.spatial_correlation_sample(a, b, kernel_size=(3, 1), patch_size=(3, 5), stride=(1, 2),
                                         padding=(1, 2), dilation=(1, 2), dilation_patch=(2, 1))
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    save("corr_rect_f32", input1=a, input2=b, output=out, grad_output=go, grad_input1=a.grad,
         grad_input2=b.grad, params=np.array([3, 1, 3, 5, 1, 2, 1, 2, 1, 2, 2, 1]))

    # full-size FlowNetC correlation: inputs are regenerated from the seed, only digests stored
    g = torch.Generator().manual_seed(18)
    a = torch.randn(1, 256, 48, 160, generator=g).requires_grad_(True)
    b = torch.randn(1, 256, 48, 160, generator=g).requires_grad_(True)
    out = scs.spatial_correlation_sample(a, b, kernel_size=1, patch_size=21, stride=1, padding=0,
                                         dilation_patch=2)
    go = torch.randn(out.shape, generator=g)
    out.backward(go)
    idx = torch.randint(0, out.numel(), (64,), generator=g)
    gidx = torch.randint(0, a.numel(), (64,), generator=g)
    save("corr_flownetc_full_digest", seed=18,
         out_sum=out.double().sum(), out_abs=out.double().abs().sum(), out_idx=idx,
         out_val=out.flatten()[idx],
         g1_sum=a.grad.double().sum(), g1_abs=a.grad.double().abs().sum(), g_idx=gidx,
         g1_val=a.grad.flatten()[gidx], g2_sum=b.grad.double().sum(),
         g2_abs=b.grad.double().abs().sum(), g2_val=b.grad.flatten()[gidx])

    # correlate() wrapper of submodules.py:124-138 (view order + divide by C)
    sub = rh.ref_module("models.submodules")
    g = torch.Generator().manual_seed(19)
    a = torch.randn(1, 8, 6, 10, generator=g)
    b = torch.randn(1, 8, 6, 10, generator=g)
    save("correlate_wrapper_f32", input1=a, input2=b, output=sub.correlate(a, b))


# ------------------------------------------------------------------------------------------ RAFT corr
def gen_raft_corr():
    """CorrBlock (models/raft/corr.py:26-106): all-pairs pyramid + lookup.  It is also the pin for
    alt_cuda_corr, whose values equal CorrBlock's."""
    corr_mod = rh.ref_module("models.raft.corr")
    g = torch.Generator().manual_seed(21)
    B, C, H, W = 2, 32, 12, 16
    f1 = torch.randn(B, C, H, W, generator=g)
    f2 = torch.randn(B, C, H, W, generator=g)
    from models.raft.utils.utils import coords_grid  # noqa
    coords = coords_grid(B, H, W) + 3.0 * torch.randn(B, 2, H, W, generator=g)
    blk = corr_mod.CorrBlock(f1, f2, num_levels=4, radius=4)
    out = blk(coords)
    save("raft_corrblock_lookup", fmap1=f1, fmap2=f2, coords=coords, output=out,
         level0=blk.corr_pyramid[0], level3=blk.corr_pyramid[3])


GENERATORS = {"ops": gen_ops, "raft_corr": gen_raft_corr}


def main(argv):
    rh.install()
    # model-level generators live next to this file and register themselves
    try:
        import make_golden_models as mgm  # noqa
        GENERATORS.update(mgm.GENERATORS)
    except ImportError:
        pass
    which = argv or list(GENERATORS)
    for name in which:
        print(f"== {name}")
        GENERATORS[name]()


if __name__ == "__main__":
    torch.set_num_threads(8)
    main(sys.argv[1:])

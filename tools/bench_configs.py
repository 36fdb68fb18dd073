"""Step times of the other BASELINE.json configs on one MI355X (not the bench.py headline line):
    C2 FlowNetC patch attack B=8 | C3 RAFT (all-pairs / alt_corr) patch attack | C4 PWC-Net patch attack B=8
    C5 FlowNet2 448x1024 universal-perturbation step
Prints one JSON object per config: ms per inner-loop iteration, pairs*iterations/s.
    python tools/bench_configs.py [c2 c3 c3alt c4 c5] [--steps K]
"""
import argparse
import json
import os
import sys
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow  # noqa: E402
from understanding_flow_robustness_amd.patch_attack import PatchAttackStep  # noqa: E402
from understanding_flow_robustness_amd.universal_perturbation import UniversalPerturbationStep  # noqa: E402

DEV = "cuda:0"


PEAK_SPLIT6_TFLOPS = 2500.0 / 6     # bf16 dense MFMA peak / six products per float32 product (csrc/igemm.hip)


def event_time(fn, iters=6, warm=1):
    """Average duration (ms) of `fn` with HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def igemm_roofline(net):
    """The dominant kernel of a config = csrc/igemm.hip: every prepared launch of every native engine the network's step ran
    on, timed live with HIP events on the step's own buffers (they hold the last iteration's real activations: zero operands
    draw less power and clock higher), algorithmic FLOPs 2 * rows * taps * Cin * Cout / the six-product ceiling."""
    engines = [eng for mod in net.modules() for attr in ("_ufr_head_engines", "_ufr_encoder_engines", "_ufr_plane_graphs")
               for eng in mod.__dict__.get(attr, {}).values()]
    batch = lambda e: getattr(e, "B", None) or getattr(e, "n", 1)
    # the step's batch = the largest `B` of a head / update / graph engine (an encoder engine's `n` counts FRAMES: 2 per pair for
    # RAFT's feature encoder).  With more than one pair per step the clean forward that makes the target ran pair by pair on its
    # own one-pair engines: those are not the step's.  (Round 4 compared `n` and `B` directly and thereby dropped RAFT's whole
    # update block at one pair, and never looked at the PlaneGraph sub-networks of FlowNet2: its C3 / C5 fractions covered the
    # encoders / the three heads only.)
    heads = [eng for mod in net.modules() for eng in mod.__dict__.get("_ufr_head_engines", {}).values() if hasattr(eng, "B")]
    top = max([e.B for e in heads] + [1])              # pairs per step (a stem PlaneGraph's `B` counts frames, like an encoder's `n`)
    tables = [e.launch_table() for e in engines if top == 1 or batch(e) >= top]
    ms = gflop = 0.0
    n = 0
    for table in tables:
        for row in table:
            launch, gf = row[-2], row[-1]
            ms += event_time(launch)
            gflop += gf
            n += 1
    if not n:
        return None
    return dict(bound="mfma", kernel="igemm (csrc/igemm.hip), every prepared launch of the config's engines, run once each",
                achieved=round(gflop / ms, 1), peak=round(PEAK_SPLIT6_TFLOPS, 1), unit="TFLOP/s", frac=round(gflop / ms / PEAK_SPLIT6_TFLOPS, 3),
                launches=n, ms=round(ms, 3), algorithmic_gflop=round(gflop, 1), traffic=None)


def timed(step, steps):
    step.enqueue(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.enqueue(steps)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / steps


def patch_case(name, flownet, seed, B, H, W, steps, **extra):
    args = Namespace(flownet=flownet, l2=False, alpha=0.0, lr=1000.0, max_count=2, **extra)
    net = fetch_model(args, synthetic_seed=seed).to(DEV)
    # float32 unless the caller opts into RAFT's reduced precision with UFR_RAFT_PRECISION=bf16 | bf16x3 (flownets/raft.py)
    args.mixed_precision = os.environ.get("UFR_RAFT_PRECISION", "fp32").lower() not in ("fp32", "float32")
    g = torch.Generator().manual_seed(0)
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    if B > 1:          # one patch in patch coordinates behind the B pairs
        mask = torch.ones(1, 3, 51, 51, device=DEV)
        patch = torch.rand(1, 3, 51, 51, generator=g).to(DEV)
        placed = dict(origins=[(100, 600)] * B)
    else:
        mask = torch.zeros(B, 3, H, W, device=DEV)
        mask[:, :, 100:151, 600:651] = 1
        patch = torch.rand(1, 3, H, W, generator=g).to(DEV)
        placed = {}
    with torch.no_grad():
        target = -torch.cat([predict_flow(net, None, tgt[i:i + 1], ref[i:i + 1], args) for i in range(B)])
    step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51) if B > 1 else None)
    step.load(tgt, ref, patch, mask, patch, target, **placed)
    step.run(0)
    ms = timed(step, steps)
    mem = torch.cuda.max_memory_allocated() / 2 ** 30
    if flownet == "RAFT":
        name += f" [{net.products()} product(s) per float32 product]" if net.products() != 6 else ""
    return dict(config=name, pairs=B, ms_per_iteration=round(ms, 3), attack_iters_per_s=round(B * 1e3 / ms, 2),
                peak_mem_gib=round(mem, 2), roofline=igemm_roofline(net))


def universal_case(steps):
    B, H, W = 1, 448, 1024
    args = Namespace(flownet="FlowNet2", n_step=10, learning_rate=2e-3, output_norm=0.02, flow_loss="cossim",
                     perturb_method="ifgsm", perturb_mode="both", add_gaussian=False)
    net = fetch_model(args, synthetic_seed=3).to(DEV)
    g = torch.Generator().manual_seed(0)
    i0, i1 = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    with torch.no_grad():
        target = -predict_flow(net, None, i0, i1, args)
    step = UniversalPerturbationStep(net, args, B, H, W, device=DEV, shared=True)
    step.load(i0, i1, torch.zeros(2, 3, H, W, device=DEV), target)
    step.run(0)
    step.run(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step.run(steps)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    return dict(config="c5 FlowNet2 448x1024 universal perturbation step", pairs=B, ms_per_iteration=round(ms, 3),
                attack_iters_per_s=round(B * 1e3 / ms, 2), peak_mem_gib=round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                roofline=igemm_roofline(net))


def measure(which, steps=10):
    torch.cuda.reset_peak_memory_stats()
    if which == "c2":
        r = patch_case("c2 FlowNetC 384x1280 patch attack", "FlowNetC", 0, 8, 384, 1280, steps)
    elif which == "c2b1":                   # the reference's own batch size (train() attacks one pair per loader item)
        r = patch_case("c2 FlowNetC 384x1280 patch attack, 1 pair (the reference's batch size)", "FlowNetC", 0, 1, 384, 1280, steps)
    elif which == "c3":
        r = patch_case("c3 RAFT 384x1280 all-pairs, 12 GRU iterations, patch attack", "RAFT", 2, 1, 384, 1280, steps)
    elif which == "c3alt":
        r = patch_case("c3 RAFT 384x1280 alt_cuda_corr, 12 GRU iterations, patch attack", "RAFT", 2, 1, 384, 1280, steps,
                       alternate_corr=True)
    elif which in ("c3b8", "c3altb8"):      # the same RAFT step with 8 pairs behind one patch (batch extension)
        r = patch_case(f"c3 RAFT 384x1280 {'alt_cuda_corr' if 'alt' in which else 'all-pairs'}, 12 GRU iterations, "
                       "patch attack, 8 pairs", "RAFT", 2, 8, 384, 1280, steps, alternate_corr="alt" in which)
    elif which == "c4":
        r = patch_case("c4 PWC-Net 384x1280 patch attack", "PWCNet", 1, 8, 384, 1280, steps)
    else:
        r = universal_case(steps)
    torch.cuda.empty_cache()
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["c2", "c3", "c3alt", "c4", "c5"])
    ap.add_argument("--steps", type=int, default=10)
    opt = ap.parse_args()
    torch.backends.cudnn.benchmark = True
    for w in opt.which:
        print(json.dumps(measure(w, opt.steps)), flush=True)


if __name__ == "__main__":
    main()

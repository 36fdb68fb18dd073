"""CPU suite, part 5: the N>1 protocol on `gloo`, world_size 2.  Patch attack: the ranks share ONE patch
in patch coordinates; each rank holds half of the frame pairs AT DIFFERENT PLACEMENTS, crops its pairs' pre-clamp
gradients back to [3,ph,pw] (the CPU oracle stands in for the HIP step), and the product's ShardedExchange all-gathers
the 31-KB-class rows [crop | loss]; every rank adds the rows in rank order and applies the clamp.  The result must equal
the single-process batch with the same summation groups BIT FOR BIT, and a patch pixel must carry both ranks'
gradients."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

PH = PW = 20
ORIGINS = [(10, 20), (30, 80), (5, 100), (40, 8)]          # four pairs, four different placements
LR = 5e4


def _rendezvous(tmp_path):
    """A file-store rendezvous in the test's own directory: no TCP port to pick, nothing to resolve."""
    return f"file://{tmp_path}/rendezvous"


def _inputs():
    g = torch.Generator().manual_seed(77)
    tgt, ref = torch.rand(4, 3, 64, 128, generator=g), torch.rand(4, 3, 64, 128, generator=g)
    yy, xx = torch.meshgrid(torch.arange(PH), torch.arange(PW), indexing="ij")
    mask_p = (((yy - 10) ** 2 + (xx - 10) ** 2) <= 64).float().expand(1, 3, PH, PW).contiguous()   # circular
    patch0 = torch.rand(1, 3, PH, PW, generator=g)
    target = torch.randn(4, 2, 64, 128, generator=g)
    return tgt, ref, mask_p, patch0, target


def _rank_main(rank, world, rdzv, out_dir):
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    from understanding_flow_robustness_amd.patch_attack import CLAMP_BOUND, ShardedExchange
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    tgt, ref, mask_p, patch0, target = _inputs()
    per = 4 // world
    sl = slice(rank * per, (rank + 1) * per)                 # this rank's shard of the batch
    tgt, ref, target, origins = tgt[sl], ref[sl], target[sl], ORIGINS[sl]
    exchange = ShardedExchange()
    assert exchange.world == world and exchange.rank == rank
    patch = patch0.clone()
    n = patch.numel()
    mask = fo.place(mask_p, origins, 64, 128)
    shown = (mask_p != 0).float()
    rows_all = torch.zeros(world, n + 1)
    first_rows = None
    for _ in range(2):
        canvas = fo.place(patch, origins, 64, 128)
        adv_t = ((1 - mask) * tgt + mask * canvas).clamp(0, 1).requires_grad_(True)
        adv_r = ((1 - mask) * ref + mask * canvas).clamp(0, 1).requires_grad_(True)
        loss = fo.flow_loss(fo.flownetc_forward(sd, adv_t, adv_r), target) / world    # PatchAttackStep: weight (1-alpha)/world
        g_t, g_r = torch.autograd.grad(loss, (adv_t, adv_r))
        row = torch.zeros_like(patch)                                                  # ufr_patch_grad_crop
        for b, (oy, ox) in enumerate(origins):
            row = row + (g_t[b:b + 1, :, oy:oy + PH, ox:ox + PW] + g_r[b:b + 1, :, oy:oy + PH, ox:ox + PW])
        rows_local = torch.cat(((row * shown).reshape(-1), loss.detach().reshape(1))).view(1, n + 1)
        exchange.gather(rows_local, rows_all)                                          # all-gather over gloo
        G = torch.zeros(n)
        for r in range(world):                                                         # ufr_patch_apply: ascending rank order
            G = G + rows_all[r, :n]
        patch = patch - torch.clamp(0.5 * LR * G.view_as(patch), -CLAMP_BOUND, CLAMP_BOUND)
        if first_rows is None:
            first_rows = rows_all.clone()
    torch.save(dict(patch=patch, loss=rows_all[:, n].sum(), rows=first_rows), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_patch_update_equals_single_process_batch(oracle, tmp_path):
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    mp.spawn(_rank_main, args=(2, _rendezvous(tmp_path), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["patch"], r1["patch"]), "ranks must hold bit-identical patches after the exchange"
    assert torch.equal(r0["loss"], r1["loss"]) and torch.equal(r0["rows"], r1["rows"])
    # a patch pixel receives BOTH ranks' gradients although no two pairs overlap on the canvas
    n = 3 * PH * PW
    both = (r0["rows"][0, :n] != 0) & (r0["rows"][1, :n] != 0)
    shown = _inputs()[2].reshape(-1) != 0
    assert float(both[shown].float().mean()) > 0.99 and not bool(both[~shown].any())
    # single process, batch of 4, the same two summation groups: bit-identical
    torch.set_num_threads(2)                                  # same oneDNN partitioning as the ranks
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    tgt, ref, mask_p, patch0, target = _inputs()
    trace = []
    patch = patch0.clone()
    # the ranks clamp their first paste (frames are in [0,1]: clamp is the identity there)
    fo.patch_attack_placed(lambda a, b: fo.flownetc_forward(sd, a, b), tgt, ref, patch, mask_p, ORIGINS, target, lr=LR,
                           max_count=2, groups=2, trace=trace)
    upd = float((patch - patch0).abs().max())
    assert 1e-3 < upd
    err = float((r0["patch"] - patch).abs().max())
    assert err <= 1e-6 * max(upd, 1.0), f"sharded vs single-process patch: {err:.3e} (update {upd:.3e})"
    assert abs(float(r0["loss"]) - trace[-1]["loss"]) < 1e-6


# ------------------------------------------------------------------ universal perturbation (config C5)
def _universal_inputs():
    g = torch.Generator().manual_seed(91)
    img0, img1 = torch.rand(2, 3, 64, 128, generator=g), torch.rand(2, 3, 64, 128, generator=g)
    target = 4.0 * torch.randn(2, 2, 64, 128, generator=g)
    return img0, img1, target


def _universal_rank_main(rank, world, rdzv, out_dir):
    """UniversalPerturbationStep's N>1 protocol (universal_perturbation.py::_part_a / _update modes 1 and 2):
    local loss scaled by 1/(B*world*H*W), local sum of the two image gradients packed as [2,3,H,W | loss],
    one all-reduce, then sign / clamp(+-eps) applied identically by every rank."""
    torch.set_num_threads(2)
    dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    from understanding_flow_robustness_amd.patch_attack import ShardedExchange
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    img0, img1, target = (x[rank:rank + 1] for x in _universal_inputs())
    exchange = ShardedExchange()
    lr, eps = 2e-3, 0.005
    delta = torch.zeros(2, 3, 64, 128)
    n = delta.numel()
    for _ in range(3):
        adv0 = torch.clamp(img0 + delta[0], 0, 1).requires_grad_(True)
        adv1 = torch.clamp(img1 + delta[1], 0, 1).requires_grad_(True)
        loss = fo.compute_flow_loss(fo.flownetc_forward(sd, adv0, adv1), target, "l2") / world
        g0, g1 = torch.autograd.grad(loss, (adv0, adv1))
        packed = torch.cat((g0.sum(0).reshape(-1), g1.sum(0).reshape(-1), loss.detach().reshape(1)))
        exchange(packed)
        g = packed[:n].view_as(delta)
        delta = torch.clamp(delta - lr * torch.sign(g), -eps, eps)
    torch.save(dict(delta=delta, loss=packed[n:].clone()), os.path.join(out_dir, f"u_rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_universal_update_equals_single_process_batch(oracle, tmp_path):
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    mp.spawn(_universal_rank_main, args=(2, _rendezvous(tmp_path), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "u_rank0.pt"), torch.load(tmp_path / "u_rank1.pt")
    assert torch.equal(r0["delta"], r1["delta"]), "ranks must hold bit-identical perturbations after the exchange"
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    img0, img1, target = _universal_inputs()
    _, _, d = fo.universal_attack(lambda a, b: fo.flownetc_forward(sd, a, b), img0, img1, torch.zeros(1, 2, 3, 64, 128),
                                  target, n_step=3, lr=2e-3, eps=0.005, flow_loss="l2", shared=True)
    # sign() of a sum computed in a different order: only pixels whose summed gradient is ~0 may flip
    differing = float((r0["delta"] != d).float().mean())
    assert differing <= 2e-3, f"{differing:.3%} of the perturbation entries differ"
    assert float((r0["delta"] - d).abs().max()) <= 2 * 2e-3 * 3


@pytest.mark.timeout(300)
def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it must start two ranks (a child torch.distributed.run on
    127.0.0.1) instead of silently running one: without a GPU each rank refuses to run -- there is no CPU fallback -- and
    the failure is relayed as a non-zero exit code, never as a result line."""
    import subprocess
    import sys
    from conftest import ROOT
    if torch.cuda.is_available():
        pytest.skip("covered by tests/test_sharding_gpu.py::test_bench_two_rank_rehearsal on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=280, cwd=ROOT)
    assert out.returncode != 0 and not [l for l in out.stdout.splitlines() if l.startswith("{")]
    # (the launcher ends the other rank as soon as one has failed: one refusal is always there, the second usually)
    assert out.stderr.count("bench.py needs an MI355X") >= 1 and "nproc-per-node=2" not in out.stdout, out.stderr[-2000:]

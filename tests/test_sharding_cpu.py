"""CPU suite, part 5: the N>1 protocol on `gloo`, world_size 2.  Each rank holds half of the frame
pairs, computes its local pre-clamp gradient sum (here with the CPU oracle standing in for the HIP
step), packs [gradient sum | loss] and calls the product's ShardedExchange; the non-linear update
applied after the all-reduce must equal the single-process batch result."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    g = torch.Generator().manual_seed(77)
    tgt, ref = torch.rand(2, 3, 64, 128, generator=g), torch.rand(2, 3, 64, 128, generator=g)
    mask = torch.zeros(2, 3, 64, 128)
    mask[0, :, 10:30, 20:40] = 1
    mask[1, :, 30:50, 80:100] = 1
    patch0 = torch.rand(1, 3, 64, 128, generator=g)
    target = torch.randn(2, 2, 64, 128, generator=g)
    return tgt, ref, mask, patch0, target


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    from understanding_flow_robustness_amd.patch_attack import CLAMP_BOUND, ShardedExchange
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    tgt, ref, mask, patch0, target = _inputs()
    sl = slice(rank, rank + 1)                       # this rank's shard of the batch
    tgt, ref, mask, target = tgt[sl], ref[sl], mask[sl], target[sl]
    exchange = ShardedExchange()
    assert exchange.world == world
    patch, lr = patch0.clone(), 5e4
    CHW = patch.numel()
    for _ in range(2):
        adv_t = ((1 - mask) * tgt + mask * patch).clamp(0, 1).requires_grad_(True)
        adv_r = ((1 - mask) * ref + mask * patch).clamp(0, 1).requires_grad_(True)
        flow = fo.flownetc_forward(sd, adv_t, adv_r)
        loss = fo.flow_loss(flow, target) / world     # PatchAttackStep: weight (1-alpha)/world
        g_t, g_r = torch.autograd.grad(loss, (adv_t, adv_r))
        gsum = ((g_t + g_r) * (mask != 0).float()).sum(0)                                 # mode 1 | MASKED_SUM
        packed = torch.cat((gsum.reshape(-1), loss.detach().reshape(1)))
        exchange(packed)                              # all-reduce(sum) over gloo
        patch = patch - torch.clamp(0.5 * lr * packed[:CHW].view_as(patch), -CLAMP_BOUND, CLAMP_BOUND)  # mode 2
    torch.save(dict(patch=patch, loss=packed[CHW:].clone()), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_patch_update_equals_single_process_batch(oracle, tmp_path):
    from oracle import flow_oracle as fo
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    port = _free_port()
    mp.spawn(_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(r0["patch"], r1["patch"]), "ranks must hold bit-identical patches after the exchange"
    assert torch.equal(r0["loss"], r1["loss"])
    # single process, batch of 2, first paste clamped like the ranks' (frames are in [0,1] anyway)
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    tgt, ref, mask, patch0, target = _inputs()
    trace = []
    patch = patch0.clone()
    fo.patch_attack(lambda a, b: fo.flownetc_forward(sd, a, b), tgt, ref, patch, mask, patch0, target, lr=5e4,
                    max_count=2, trace=trace)
    upd = float((patch - patch0).abs().max())
    assert float((r0["patch"] - patch).abs().max()) <= 1e-5 * max(upd, 1.0)
    assert abs(float(r0["loss"]) - trace[-1]["loss"]) < 1e-6


# ------------------------------------------------------------------ universal perturbation (config C5)
def _universal_inputs():
    g = torch.Generator().manual_seed(91)
    img0, img1 = torch.rand(2, 3, 64, 128, generator=g), torch.rand(2, 3, 64, 128, generator=g)
    target = 4.0 * torch.randn(2, 2, 64, 128, generator=g)
    return img0, img1, target


def _universal_rank_main(rank, world, port, out_dir):
    """UniversalPerturbationStep's N>1 protocol (universal_perturbation.py::_part_a / _update modes 1 and 2):
    local loss scaled by 1/(B*world*H*W), local sum of the two image gradients packed as [2,3,H,W | loss],
    one all-reduce, then sign / clamp(+-eps) applied identically by every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
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
    port = _free_port()
    mp.spawn(_universal_rank_main, args=(2, port, str(tmp_path)), nprocs=2, join=True)
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

"""GPU suite: the N>1 path of the fused step on real kernels.  Two ranks hold two frame pairs each, at
four different placements, behind ONE patch in patch coordinates: part-A graph -> all-gather of the [crop | loss] rows
-> part-B graph.  With two visible devices the ranks take one each and the exchange runs over RCCL (backend "nccl");
on the one-GPU test box they share the device and gloo moves the rows.  Both ranks must end with bit-identical patches
that equal the single-process step run with the same two summation groups."""
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PH = PW = 20
ORIGINS = [(10, 20), (30, 80), (5, 100), (40, 8)]


def _rendezvous(tmp_path):
    """A file-store rendezvous in the test's own directory: no TCP port to pick (bind-then-close races with every other
    process on the box), nothing to resolve."""
    return f"file://{tmp_path}/rendezvous"


def _inputs():
    g = torch.Generator().manual_seed(5)
    tgt, ref = torch.rand(4, 3, 64, 128, generator=g), torch.rand(4, 3, 64, 128, generator=g)
    yy, xx = torch.meshgrid(torch.arange(PH), torch.arange(PW), indexing="ij")
    mask_p = (((yy - 10) ** 2 + (xx - 10) ** 2) <= 64).float().expand(1, 3, PH, PW).contiguous()
    patch0 = torch.rand(1, 3, PH, PW, generator=g)
    target = torch.randn(4, 2, 64, 128, generator=g)
    return tgt, ref, mask_p, patch0, target


def _make_step(batch, exchange, dev, groups=1, flownet="FlowNetC"):
    from argparse import Namespace
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    args = Namespace(flownet=flownet, l2=False, alpha=0.0, lr=5e4, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(dev)
    return PatchAttackStep(net, args, batch, 64, 128, device=dev, exchange=exchange, patch_hw=(PH, PW), sum_groups=groups)


def _init_ranks(rank, world, rdzv):
    """One device per rank over RCCL when the box has them; on the one-GPU test box the ranks share cuda:0 and gloo moves the
    rows (the product code is the same: ShardedExchange calls torch.distributed)."""
    sys.path.insert(0, ROOT)
    two_devices = torch.cuda.device_count() >= world
    dev = f"cuda:{rank}" if two_devices else DEV
    torch.cuda.set_device(dev)
    if two_devices:                                    # RCCL over xGMI: the production transport
        dist.init_process_group("nccl", init_method=rdzv, rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        dist.init_process_group("gloo", init_method=rdzv, rank=rank, world_size=world)
    return dev


def _rank_main(rank, world, rdzv, out_dir, flownet="FlowNetC"):
    dev = _init_ranks(rank, world, rdzv)
    from understanding_flow_robustness_amd.patch_attack import ShardedExchange
    tgt, ref, mask_p, patch0, target = _inputs()
    sl = slice(2 * rank, 2 * rank + 2)
    step = _make_step(2, ShardedExchange(), dev, flownet=flownet)
    step.load(tgt[sl].to(dev), ref[sl].to(dev), patch0.to(dev), mask_p.to(dev), patch0.to(dev), target[sl].to(dev),
              origins=ORIGINS[sl])
    n, loss = step.run(2)
    torch.save(dict(patch=step.patch.cpu(), n=n, loss=loss, graphs=(step.graph is not None, step.graph_b is not None),
                    backend=dist.get_backend(), rows=step.rows_all.cpu()), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("flownet", ["FlowNetC", "PWCNet"])
def test_two_ranks_match_single_process_batch(tmp_path, flownet):
    """FlowNetC = config C2's step, PWCNet = config C4's (PWC-Net behind the exchange: SURVEY.md 8e, "C4: 64 -> 8 x 8"):
    the product's own sharded PatchAttackStep on two ranks against its single-process form with the same two summation
    groups."""
    mp.spawn(_rank_main, args=(2, _rendezvous(tmp_path), str(tmp_path), flownet), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["graphs"] == (True, True) and r0["n"] == r1["n"] == 2
    assert r0["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert torch.equal(r0["patch"], r1["patch"]) and torch.equal(r0["rows"], r1["rows"])
    n = 3 * PH * PW
    shown = _inputs()[1 + 1].reshape(-1) != 0
    both = (r0["rows"][0, :n] != 0) & (r0["rows"][1, :n] != 0)     # a patch pixel carries both ranks' gradients
    assert float(both[shown].float().mean()) > 0.99
    tgt, ref, mask_p, patch0, target = _inputs()
    step = _make_step(4, None, DEV, groups=2, flownet=flownet)              # same summation tree as the two ranks
    step.load(tgt.to(DEV), ref.to(DEV), patch0.to(DEV), mask_p.to(DEV), patch0.to(DEV), target.to(DEV), origins=ORIGINS)
    n_it, loss = step.run(2)
    upd = float((step.patch.cpu() - patch0).abs().max())
    assert n_it == 2 and upd > 1e-3
    # the convolutions of a batch of 2 and of a batch of 4 may take different MIOpen kernels: 1e-4 of the update
    assert float((step.patch.cpu() - r0["patch"]).abs().max()) <= 1e-4 * max(upd, 1.0)
    assert abs(loss - r0["loss"]) < 1e-5


# ------------------------------------------------------------------ universal perturbation (config C5's step)
U_LR, U_EPS, U_STEPS = 2e-3, 0.005, 3


def _universal_inputs(gt_channels):
    g = torch.Generator().manual_seed(91)
    img0, img1 = torch.rand(2, 3, 64, 128, generator=g), torch.rand(2, 3, 64, 128, generator=g)
    target = 4.0 * torch.randn(2, gt_channels, 64, 128, generator=g)
    if gt_channels == 3:                                   # KITTI-style validity channel (perturb_model.py:141-143)
        target[:, 2] = (torch.rand(2, 64, 128, generator=g) > 0.3).float()
    return img0, img1, target


def _universal_delta0(flownet):
    """FlowNetC starts from zeros like the reference (universal_perturbation.py:314); FlowNet2's four floor() warps make its
    gradient piecewise, so its case starts from a perturbation "as after earlier samples" (as the C5 test does) where the
    gradient is well away from zero."""
    if flownet != "FlowNet2":
        return torch.zeros(2, 3, 64, 128)
    return (torch.rand(2, 3, 64, 128, generator=torch.Generator().manual_seed(92)) * 2 - 1) * 0.5 * U_EPS


def _universal_step(batch, exchange, dev, flow_loss, gt_channels, flownet="FlowNetC"):
    from argparse import Namespace
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.universal_perturbation import UniversalPerturbationStep
    args = Namespace(flownet=flownet, n_step=U_STEPS, learning_rate=U_LR, output_norm=U_EPS, flow_loss=flow_loss,
                     perturb_method="ifgsm", perturb_mode="both", add_gaussian=False)
    net = fetch_model(args, synthetic_seed=3 if flownet == "FlowNet2" else 0).to(dev)
    return UniversalPerturbationStep(net, args, batch, 64, 128, gt_channels=gt_channels, device=dev, shared=True,
                                     exchange=exchange)


def _universal_rank_main(rank, world, rdzv, out_dir, flow_loss, gt_channels, flownet):
    dev = _init_ranks(rank, world, rdzv)
    from understanding_flow_robustness_amd.patch_attack import ShardedExchange
    img0, img1, target = (x[rank:rank + 1].to(dev) for x in _universal_inputs(gt_channels))
    step = _universal_step(1, ShardedExchange(), dev, flow_loss, gt_channels, flownet)
    step.load(img0, img1, _universal_delta0(flownet).to(dev), target)
    step.run(U_STEPS)
    torch.cuda.synchronize()
    torch.save(dict(delta=step.delta.cpu(), loss=float(step.loss_cur), scale=float(step.scale_t),
                    graphs=(step.graph is not None, step.graph_b is not None), backend=dist.get_backend()),
               os.path.join(out_dir, f"u_rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("flow_loss,gt_channels,flownet", [("cossim", 2, "FlowNetC"), ("l2", 3, "FlowNetC"),
                                                           ("cossim", 2, "FlowNet2")])
def test_two_rank_universal_step_matches_single_process(tmp_path, flow_loss, gt_channels, flownet):
    """The product's own sharded UniversalPerturbationStep (universal_perturbation.py `world > 1`: `_update` modes 1 / 2
    around the packed all-reduce, the `valid` all-reduce of a 3-channel target) on two ranks of one pair each: bit-identical
    perturbations on both ranks, and the single-process `shared=True` step over the same two pairs up to sign flips where the
    summed gradient is ~0 ("universal: [2,3,H,W] fp32 all-reduced, then sign").  The FlowNet2 case is config
    C5's own network (Correlation + 4 Resample2d + 6 ChannelNorm, models/flownet2_models.py:122-205) behind the exchange."""
    mp.spawn(_universal_rank_main, args=(2, _rendezvous(tmp_path), str(tmp_path), flow_loss, gt_channels, flownet), nprocs=2,
             join=True)
    r0, r1 = torch.load(tmp_path / "u_rank0.pt"), torch.load(tmp_path / "u_rank1.pt")
    assert r0["graphs"] == (True, True)
    assert r0["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert torch.equal(r0["delta"], r1["delta"]), "ranks must hold bit-identical perturbations after the exchange"
    assert r0["loss"] == r1["loss"] and r0["scale"] == r1["scale"]
    img0, img1, target = (x.to(DEV) for x in _universal_inputs(gt_channels))
    step = _universal_step(2, None, DEV, flow_loss, gt_channels, flownet)
    step.load(img0, img1, _universal_delta0(flownet).to(DEV), target)
    step.run(U_STEPS)
    d = step.delta.cpu()
    assert float(d.abs().max()) > 0.5 * U_EPS                       # the steps did move the perturbation
    assert abs(float(step.scale_t) - r0["scale"]) <= 1e-6 * r0["scale"]
    differing = float((r0["delta"] != d).float().mean())
    assert differing <= 2e-3, f"{differing:.3%} of the perturbation entries differ"
    assert float((r0["delta"] - d).abs().max()) <= 2 * U_LR * U_STEPS
    assert abs(float(step.loss_cur) - r0["loss"]) <= 1e-4 * max(1.0, abs(r0["loss"]))


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("ranks,config", [(2, "c2"), (4, "c2"), (2, "c4"), (2, "c5")])
def test_bench_two_rank_rehearsal(ranks, config):
    """`python bench.py --gpus N` WITHOUT a launcher: bench.py starts the N ranks itself (a child torch.distributed.run,
    before it touches the GPU), relays rank 0's one JSON line and reports n_gpus = N; whole-job value = N ranks x 8 pairs x
    steps / time.  On the one-GPU box the ranks share the device and gloo moves the rows (UFR_DIST_BACKEND); four ranks + this
    process stay inside the box's limit of six GPU processes."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None), env.pop("RANK", None), env.pop("LOCAL_RANK", None)
    if torch.cuda.device_count() < 2:
        env["UFR_DIST_BACKEND"] = "gloo"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1", "--config", config]
    pairs = 1 if config == "c5" else 8
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == ranks and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert abs(rec["value"] - ranks * pairs * 3 / (rec["ms_per_step"] * 3 / 1e3)) < 0.02 * rec["value"]
    assert rec["config"]["global_pairs"] == ranks * pairs and {"c2": "FlowNetC", "c4": "PWC-Net", "c5": "FlowNet2"}[config] in rec["config"]["workload"]
    if ranks > 2 or config != "c2":
        return
    # a launcher world that disagrees with --gpus is refused instead of printing a line for the wrong job
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=dict(env, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert bad.returncode != 0 and "WORLD_SIZE" in (bad.stderr + bad.stdout)


def _rccl_rank(rank, rdzv, out_dir):
    os.environ.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    torch.cuda.set_device(DEV)
    dist.init_process_group("nccl", init_method=rdzv, rank=0, world_size=1, device_id=torch.device(DEV))
    rows_local = torch.arange(2 * 7, dtype=torch.float32, device=DEV).view(2, 7)
    rows_all = torch.full((2, 7), -1.0, device=DEV)
    dist.all_gather_into_tensor(rows_all, rows_local)              # the call ShardedExchange.gather makes for world > 1
    packed = torch.ones(1 << 20, device=DEV)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM)                  # ShardedExchange.__call__ (universal perturbation)
    dist.barrier()
    torch.cuda.synchronize()
    torch.save(dict(backend=dist.get_backend(), gathered=rows_all.cpu(), reduced=float(packed.sum())),
               os.path.join(out_dir, "rccl.pt"))
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_initialises_and_runs_the_exchange_collectives(tmp_path):
    """The one-GPU box cannot run two RCCL ranks, so the N > 1 tests above move their rows over gloo.  This one brings RCCL
    itself up on the MI355X (backend "nccl", one rank) and issues the two collectives the exchange uses on HIP tensors."""
    mp.spawn(_rccl_rank, args=(_rendezvous(tmp_path), str(tmp_path)), nprocs=1, join=True)
    r = torch.load(tmp_path / "rccl.pt")
    assert r["backend"] == "nccl"
    assert torch.equal(r["gathered"], torch.arange(14, dtype=torch.float32).view(2, 7)) and r["reduced"] == float(1 << 20)

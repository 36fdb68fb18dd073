"""GPU suite: the N>1 path of the fused step on real kernels.  Two ranks share the single device of
the test box (gloo moves the packed gradient; on a multi-GPU node the same code runs over RCCL):
part-A graph -> all-reduce of [gradient sum | loss] -> part-B graph must reproduce the
single-process batch-2 step, and both ranks must end with bit-identical patches."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _inputs():
    g = torch.Generator().manual_seed(5)
    tgt, ref = torch.rand(2, 3, 64, 128, generator=g), torch.rand(2, 3, 64, 128, generator=g)
    mask = torch.zeros(2, 3, 64, 128)
    mask[0, :, 10:30, 20:40] = 1
    mask[1, :, 30:50, 80:100] = 1
    patch0 = torch.rand(1, 3, 64, 128, generator=g)
    target = torch.randn(2, 2, 64, 128, generator=g)
    return tgt, ref, mask, patch0, target


def _make_step(batch, exchange):
    from argparse import Namespace
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=5e4, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    return PatchAttackStep(net, args, batch, 64, 128, device=DEV, exchange=exchange)


def _rank_main(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from understanding_flow_robustness_amd.patch_attack import ShardedExchange
    tgt, ref, mask, patch0, target = _inputs()
    sl = slice(rank, rank + 1)
    step = _make_step(1, ShardedExchange())
    step.load(tgt[sl].to(DEV), ref[sl].to(DEV), patch0.to(DEV), mask[sl].to(DEV), patch0.to(DEV), target[sl].to(DEV))
    n, loss = step.run(2)
    torch.save(dict(patch=step.patch.cpu(), n=n, loss=loss, graphs=(step.graph is not None, step.graph_b is not None)),
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
def test_two_ranks_match_single_process_batch(tmp_path):
    mp.spawn(_rank_main, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert r0["graphs"] == (True, True) and r0["n"] == r1["n"] == 2
    assert torch.equal(r0["patch"], r1["patch"])
    tgt, ref, mask, patch0, target = _inputs()
    step = _make_step(2, None)
    step.load(tgt.to(DEV), ref.to(DEV), patch0.to(DEV), mask.to(DEV), patch0.to(DEV), target.to(DEV))
    n, loss = step.run(2)
    upd = float((step.patch.cpu() - patch0).abs().max())
    assert float((step.patch.cpu() - r0["patch"]).abs().max()) <= 1e-4 * max(upd, 1.0)
    assert abs(loss - r0["loss"]) < 1e-5


@pytest.mark.timeout(1200)
def test_bench_two_rank_rehearsal():
    """bench.py under torch.distributed.run with 2 ranks on the one device (gloo): one JSON line,
    whole-job value = 2 ranks x 8 pairs x steps / time."""
    env = dict(os.environ, UFR_DIST_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps",
           "3", "--warmup", "1"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1100, cwd=ROOT)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak"
    assert abs(rec["value"] - 2 * 8 * 3 / (rec["ms_per_step"] * 3 / 1e3)) < 0.02 * rec["value"]

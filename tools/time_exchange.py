"""The eager exchange between the two graphs of a sharded step, timed with RCCL on ONE rank: the one piece of the
multi-GPU path with no timing at all.  One MI355X cannot run two RCCL ranks, so this measures the collective's fixed cost -- launch
path, RCCL's own kernel, the host-side enqueue between two graph replays -- not the xGMI transfer:

    all_gather_into_tensor of [1, 3*51*51 + 1] float32   (config C2 / C4: the patch-coordinate gradient rows, 31 KB per rank)
    all_reduce(sum) of 2*3*448*1024 + 1 float32           (config C5: the image-sized gradient of the universal perturbation, 11 MiB)

each (a) back to back, device time per call from HIP events, and (b) the way the step issues it: graph replay -> collective -> graph
replay, wall time per iteration against the same loop without the collective.  One JSON line per measurement.

    python tools/time_exchange.py
The first SCALE run of the driver can be read against these: per step, N ranks add the transfer (31 KB x (N-1) per rank for the
gather; 2 x 11 MiB x (N-1)/N per rank for a ring all-reduce at ~153 GB/s per xGMI link: ~0.13 ms at N = 8) to the fixed cost here.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DEV = "cuda:0"


def main():
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(DEV)
    rdzv = f"file:///tmp/ufr_exchange_rdzv_{os.getpid()}"
    dist.init_process_group("nccl", init_method=rdzv, rank=0, world_size=1, device_id=torch.device(DEV))
    from understanding_flow_robustness_amd.patch_attack import ShardedExchange
    ex = ShardedExchange()
    ex.world = 2                     # issue the collectives as a 2-rank job would (the process group itself has one rank)
    n_rows = 3 * 51 * 51 + 1
    rows_local, rows_all = torch.zeros(1, n_rows, device=DEV), torch.zeros(1, n_rows, device=DEV)
    packed = torch.zeros(2 * 3 * 448 * 1024 + 1, device=DEV)
    cases = (("all_gather_into_tensor [1, 7804] f32 (C2 / C4 rows, 31 KB)", lambda: ex.gather(rows_local, rows_all), rows_local.numel() * 4),
             ("all_reduce(sum) 2 x 3 x 448 x 1024 + 1 f32 (C5 gradient, 11 MiB)", lambda: ex(packed), packed.numel() * 4))
    # a stand-in for the step's two graphs: a few microseconds of captured work each
    buf = torch.zeros(1 << 16, device=DEV)
    graphs = []
    for _ in range(2):
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            buf.add_(1.0)
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            buf.add_(1.0)
        graphs.append(g)
    for name, call, nbytes in cases:
        for _ in range(10):
            call()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 200
        s.record()
        for _ in range(iters):
            call()
        e.record()
        e.synchronize()
        device_us = s.elapsed_time(e) / iters * 1e3

        def loop(with_collective):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                graphs[0].replay()
                if with_collective:
                    call()
                graphs[1].replay()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e6

        loop(True)
        with_us, without_us = loop(True), loop(False)
        print(json.dumps(dict(collective=name, bytes=nbytes, backend=dist.get_backend(), ranks=1,
                              device_us_per_call_back_to_back=round(device_us, 2),
                              wall_us_per_iteration_graph_collective_graph=round(with_us, 2),
                              wall_us_per_iteration_graph_graph=round(without_us, 2),
                              added_by_the_collective_us=round(with_us - without_us, 2))), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

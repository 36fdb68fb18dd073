"""Every prepared igemm launch of the PWC-Net head engine (pwc_engine.py) at config C4's size, timed with HIP events:
one JSON line per launch (ms, algorithmic GFLOP, fraction of the six-product ceiling) and the totals.
    python tools/bench_pwc_layers.py [--pairs 8]"""
import argparse
import json
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from understanding_flow_robustness_amd.flownets.utils_model import fetch_model  # noqa: E402
from understanding_flow_robustness_amd.pwc_engine import get_engine  # noqa: E402

PEAK = 2500.0 / 6


def event_time(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=8)
    ap.add_argument("--round-robin", action="store_true", help="time every launch inside a pass over ALL launches (cold operands, as in the step)")
    opt = ap.parse_args()
    net = fetch_model(Namespace(flownet="PWCNet"), synthetic_seed=1).to("cuda:0")
    for p in net.parameters():
        p.requires_grad_(False)
    eng = get_engine(net.eval(), opt.pairs, 384, 1280, "cuda:0")
    tot_ms = tot_gf = 0.0
    if opt.round_robin:
        table = eng.launch_table()
        passes = 40 if os.environ.get("PWC_LAYERS_SUSTAINED") else 6
        ev = [[(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in table] for _ in range(passes)]
        for p in range(passes):
            for i, (_, _, launch, _) in enumerate(table):
                ev[p][i][0].record()
                launch()
                ev[p][i][1].record()
        torch.cuda.synchronize()
        # sustained load: the sum over a pass as the passes go by (a burst clock at the start would show here)
        print(json.dumps(dict(pass_ms=[round(sum(a.elapsed_time(b) for a, b in ev[p]), 2) for p in range(passes)])), flush=True)
        for i, (name, kind, launch, gflop) in enumerate(table):
            ms = sum(ev[p][i][0].elapsed_time(ev[p][i][1]) for p in range(passes // 2, passes)) / (passes - passes // 2)
            d = launch.desc
            tot_ms, tot_gf = tot_ms + ms, tot_gf + gflop
            print(json.dumps(dict(launch=name, dir=kind, M=d.B * d.Hr * d.Wr, KC=d.KC, taps=d.phase[0].ntaps, N=d.N, Npad=d.Npad, splitk=d.splitk,
                                  variant=d.variant, ms=round(ms, 4), gflop=round(gflop, 2), frac=round(gflop / ms / PEAK, 3))), flush=True)
        print(json.dumps(dict(total_ms=round(tot_ms, 3), total_gflop=round(tot_gf, 1), frac=round(tot_gf / tot_ms / PEAK, 3), mode="round-robin")))
        return
    for name, kind, launch, gflop in eng.launch_table():
        d = launch.desc
        ms = event_time(launch)
        tot_ms, tot_gf = tot_ms + ms, tot_gf + gflop
        print(json.dumps(dict(launch=name, dir=kind, M=d.B * d.Hr * d.Wr, KC=d.KC, taps=d.phase[0].ntaps, N=d.N, Npad=d.Npad, splitk=d.splitk,
                              variant=d.variant, ms=round(ms, 4), gflop=round(gflop, 2), frac=round(gflop / ms / PEAK, 3))), flush=True)
    print(json.dumps(dict(total_ms=round(tot_ms, 3), total_gflop=round(tot_gf, 1), frac=round(tot_gf / tot_ms / PEAK, 3))))


if __name__ == "__main__":
    main()

"""Per-launch sweep of the igemm's kernel form and split-K factor for a whole config.

    python tools/sweep_engine_launches.py CONFIG [CONFIG ...] [--out DIR] [--keep-table]

CONFIG: c3alt | c3 (RAFT at one pair, alt_cuda_corr / all-pairs), c3altb8 (8 pairs), c5 (FlowNet2 448x1024 universal step), c4 (PWC-Net, 8 pairs),
c2b1 (FlowNetC at one pair), c2 (the headline: FlowNetC, 8 pairs; its band / window launches are skipped).  The config's step is built with `igemm.RECORD` switched on, so EVERY `make_launch` of every engine
the step uses (update block, encoders, PlaneGraph sub-networks, the FlowNetC / FlowNetS heads) is captured with its operands; the
step runs twice (the buffers then hold real activations), and every DISTINCT launch (igemm.launch_signature) is rebuilt with every
(variant, split-K) pair and timed with HIP events on those buffers, the engine's own choice first.

One JSON line per (launch, variant, splitk); per launch a summary line; at the end the table of winners that beat the engine's
rule of thumb by more than 3 %, in the format of understanding_flow_robustness_amd/igemm_tuning.json (written to
DIR/tuning_CONFIG.json with --out).  A launch whose split-K slabs are added by its consumer (no_reduce) is charged the
consumer's extra slab reads (S x M x Npad x 4 B at 3 TB/s).  Band / window launches (row_band, in_band) are skipped: the
headline's were swept in round 4 (tools/sweep_igemm_launches.py).
"""
import argparse
import json
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

DEV = "cuda:0"
CEIL = 2500.0 / 6
SPLITS = (1, 2, 3, 4, 6, 8, 12, 16, 24, 32)


OPTS = {}
COLD = [None]           # a 1 GiB tensor rewritten between timed launches (--cold): the launch then finds its operands where the
                        # step's previous kernels left them -- in HBM -- not in the L2 / MALL its own previous run filled


def timed(fn, iters=20):
    for _ in range(3):
        fn()
    if COLD[0] is not None:
        # back-to-back repeats of ONE launch keep its weights and activations in the 4 MB L2s / 256 MB MALL; inside a step 200
        # other kernels run in between.  gpurun r5_call3 / r5_call4: the hot sweep promised c3alt -0.93 ms, the step gained 0.15.
        pairs = []
        for _ in range(iters):
            COLD[0].add_(1.0)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            fn()
            e.record()
            pairs.append((s, e))
        torch.cuda.synchronize()
        return sum(s.elapsed_time(e) for s, e in pairs) / iters
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


def build_step(config):
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    from understanding_flow_robustness_amd.universal_perturbation import UniversalPerturbationStep
    g = torch.Generator().manual_seed(0)
    if config == "c5":
        B, H, W = 1, 448, 1024
        args = Namespace(flownet="FlowNet2", n_step=10, learning_rate=2e-3, output_norm=0.02, flow_loss="cossim",
                         perturb_method="ifgsm", perturb_mode="both", add_gaussian=False)
        net = fetch_model(args, synthetic_seed=3).to(DEV)
        i0, i1 = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
        with torch.no_grad():
            target = -predict_flow(net, None, i0, i1, args)
        step = UniversalPerturbationStep(net, args, B, H, W, device=DEV, shared=True)
        step.load(i0, i1, torch.zeros(2, 3, H, W, device=DEV), target)
        step.run(2)
        return step
    flownet, seed, B, extra = {"c3alt": ("RAFT", 2, 1, dict(alternate_corr=True)), "c3": ("RAFT", 2, 1, {}),
                               "c3altb8": ("RAFT", 2, 8, dict(alternate_corr=True)),
                               "c4": ("PWCNet", 1, 8, {}), "c2b1": ("FlowNetC", 0, 1, {}), "c2": ("FlowNetC", 0, 8, {})}[config]
    H, W = 384, 1280
    args = Namespace(flownet=flownet, l2=False, alpha=0.0, lr=1000.0, max_count=2, **extra)
    net = fetch_model(args, synthetic_seed=seed).to(DEV)
    args.mixed_precision = False
    tgt, ref = torch.rand(B, 3, H, W, generator=g).to(DEV), torch.rand(B, 3, H, W, generator=g).to(DEV)
    if B > 1:
        mask, patch = torch.ones(1, 3, 51, 51, device=DEV), torch.rand(1, 3, 51, 51, generator=g).to(DEV)
        placed = dict(origins=[(100, 600)] * B)
    else:
        mask = torch.zeros(B, 3, H, W, device=DEV)
        mask[:, :, 100:151, 600:651] = 1
        patch, placed = torch.rand(1, 3, H, W, generator=g).to(DEV), {}
    with torch.no_grad():
        target = -torch.cat([predict_flow(net, None, tgt[i:i + 1], ref[i:i + 1], args) for i in range(B)])
    step = PatchAttackStep(net, args, B, H, W, device=DEV, patch_hw=(51, 51) if B > 1 else None)
    step.load(tgt, ref, patch, mask, patch, target, **placed)
    step.run(2)
    return step


def sweep(config, out_dir):
    from understanding_flow_robustness_amd import igemm as ig
    ig.RECORD = []
    step = build_step(config)
    torch.cuda.synchronize()
    records, ig.RECORD = ig.RECORD, None
    seen, winners, total_now, total_best = {}, {}, 0.0, 0.0
    for r in records:
        kw = r["kw"]
        if kw.get("row_band") is not None or kw.get("in_band") is not None:
            continue
        M = r["x"].B * r["rows"][0] * r["rows"][1]
        sig = ig.launch_signature(r["wi"], M, kw, r["rows"])     # (the long form: with the row grid)
        if sig in seen:
            seen[sig]["count"] += 1
            continue
        seen[sig] = dict(r, M=M, count=1)
    # the largest batch of each signature family is the step's (the clean forward runs pair by pair on its own engines): keep all,
    # the table is keyed by M anyway
    for sig, r in seen.items():
        wi, kw, M = r["wi"], dict(r["kw"]), r["M"]
        gflop = wi.flops(M) / 1e9
        chosen = (r["variant"] if r["variant"] else 2, r["splitk"])
        variants = (6, 7, 5, 4, 2, 8) if wi.Npad % 128 == 0 else (2, 7, 8)
        if OPTS.get("variants"):                                # --variants 8: only these forms (beside the engine's own choice)
            variants = tuple(v for v in variants if v in OPTS["variants"] or v == chosen[0])
        if OPTS.get("max_n") and wi.N > OPTS["max_n"]:          # --max-n 64: only the narrow launches
            continue
        kt = max(len(t) for _, _, t in wi.phases) * wi.KC
        res = {}
        for v in variants:
            for S in SPLITS:
                if S > 1 and (v == 8 or kt // S < 2 or len(wi.phases) * S * M * wi.Npad * 4 > (2 << 30)):     # (the direct form does not split)
                    continue
                ws = torch.empty(len(wi.phases) * S * M * wi.Npad, device=DEV) if S > 1 else None
                try:
                    launch = ig.make_launch(wi, r["x"], r["in_chunk0"], r["rows"], r["out_hw"], splitk=S, ws=ws, variant=v, **kw)
                    ms = timed(launch)
                except RuntimeError as exc:
                    print(json.dumps(dict(config=config, launch=sig, variant=v, splitk=S, error=str(exc)[:120])), flush=True)
                    continue
                if kw.get("no_reduce") and S > 1:
                    ms += S * M * wi.Npad * 4 / 3e9                      # the consumer reads S slabs instead of one result
                res[(v, S)] = ms
                print(json.dumps(dict(config=config, launch=sig, variant=v, splitk=S, ms=round(ms, 4), frac=round(gflop / ms / CEIL, 3),
                                      chosen=(v, S) == chosen)), flush=True)
                del ws, launch
        if not res:
            continue
        best = min(res, key=res.get)
        cur = res.get(chosen)
        print(json.dumps(dict(config=config, launch=sig, count=r["count"], gflop=round(gflop, 3),
                              best=dict(variant=best[0], splitk=best[1], ms=round(res[best], 4), frac=round(gflop / res[best] / CEIL, 3)),
                              engine_choice=dict(variant=chosen[0], splitk=chosen[1], ms=round(cur, 4) if cur else None))), flush=True)
        if cur is not None:
            total_now += cur * r["count"]
            total_best += min(res[best], cur) * r["count"]
            if res[best] < 0.97 * cur:
                winners[sig] = [best[0], best[1]]
    print(json.dumps(dict(config=config, distinct_launches=len(seen), ms_engine_choice=round(total_now, 3), ms_best=round(total_best, 3),
                          tuning=winners)), flush=True)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
        with open(os.path.join(out_dir, f"tuning_{config}.json"), "w") as f:
            json.dump(winners, f, indent=1, sort_keys=True)
    del step
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="+")
    ap.add_argument("--out", default=None)
    ap.add_argument("--keep-table", action="store_true", help="leave igemm_tuning.json in force (default: sweep against the rule of thumb)")
    ap.add_argument("--cold", action="store_true", help="rewrite 1 GiB between timed launches: operands come from HBM, as inside a step")
    ap.add_argument("--variants", default="", help="comma / plus separated kernel forms to sweep beside the engine's choice (default: all)")
    ap.add_argument("--max-n", type=int, default=0, help="only launches with at most this many output columns")
    opt = ap.parse_args()
    if opt.cold:
        COLD[0] = torch.zeros(1 << 28, device=DEV)
    if not opt.keep_table:
        os.environ["UFR_IGEMM_TUNING"] = "0"
    OPTS.update(variants=[int(v) for v in opt.variants.replace("+", ",").split(",") if v], max_n=opt.max_n)
    for c in opt.configs:
        sweep(c, opt.out)


if __name__ == "__main__":
    main()

"""What the igemm's epilogue costs inside a launch (measured instead of priced): every prepared launch of the headline
step's engine (FlowNetC 384x1280, 8 pairs) timed with HIP events, in a process WITH and a process WITHOUT its epilogue
(UFR_IGEMM_DEBUG_SKIP_EPILOGUE=1: csrc/igemm.hip drops the accumulators after the K loop -- measurement only, the results are garbage).
The difference is the upper bound of what a persistent tile loop could return by running workgroup i's epilogue under workgroup i + 1's
K loop.  Run:  python tools/measure_epilogue_share.py > with.jsonl;  UFR_IGEMM_DEBUG_SKIP_EPILOGUE=1 python tools/measure_epilogue_share.py > without.jsonl
"""
import json
import os
import sys
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

DEV = "cuda:0"


def main():
    import bench
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    skip = os.environ.get("UFR_IGEMM_DEBUG_SKIP_EPILOGUE", "0")
    os.environ.pop("UFR_IGEMM_DEBUG_SKIP_EPILOGUE", None) if False else None
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1000.0, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    B, H, W, P = bench.B_PER_GPU, bench.H, bench.W, bench.PATCH
    step = PatchAttackStep(net, args, B, H, W, device=DEV, shared_patch=True, patch_hw=(P, P))
    tgt, ref, origins = bench.synthetic_batch(B, 1000, DEV)
    g = torch.Generator().manual_seed(7)
    patch0 = torch.rand(1, 3, P, P, generator=g).to(DEV)
    mask_p = bench.circle_mask(P).expand(1, 3, P, P).contiguous().to(DEV)
    target = torch.randn(B, 2, H, W, generator=g).to(DEV)
    step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
    step.run(2)
    eng = step.eng
    # operands with real magnitudes in both processes (without the epilogue nothing writes them): random planes
    total = {"ms": 0.0, "gflop": 0.0}
    for name, kind, tag, launch, gflop in eng.launch_table():
        t = bench.event_time(launch, 20)
        total["ms"] += t
        total["gflop"] += gflop
        print(json.dumps(dict(launch=f"{name} {kind} ({tag})", variant=launch.desc.variant, splitk=launch.desc.splitk, ms=round(t, 4),
                              gflop=round(gflop, 2), epilogue={"0": "on", "1": "skipped", "2": "computed, plane stores dropped"}.get(skip, skip))), flush=True)
    print(json.dumps(dict(launch="TOTAL over the launch table", ms=round(total["ms"], 4), gflop=round(total["gflop"], 1),
                          epilogue={"0": "on", "1": "skipped", "2": "computed, plane stores dropped"}.get(skip, skip))), flush=True)


if __name__ == "__main__":
    main()

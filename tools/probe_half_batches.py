"""Does running the headline step as TWO independent half-batches on two streams beat one batch of eight?  (Round 6: the igemm's epilogue
is 15 - 25 % of a forward launch and every workgroup of a round reaches it at the same moment, profiles/r6_igemm_epilogue_share.jsonl;
two independent launch chains let one chain's epilogues and tail rounds run under the other chain's K loops.)
    python tools/probe_half_batches.py
Builds one PatchAttackStep of 8 pairs and two of 4 pairs (the second on a deep copy of the network: its own engines and buffers), and
times N iterations of (a) the 8-pair step, (b) the two 4-pair steps back to back on one stream, (c) the two 4-pair steps on two streams."""
import copy
import json
import os
import sys
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

DEV = "cuda:0"


def main():
    import bench
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1000.0, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    net2 = copy.deepcopy(net)
    H, W, P = bench.H, bench.W, bench.PATCH
    g = torch.Generator().manual_seed(7)
    patch0 = torch.rand(1, 3, P, P, generator=g).to(DEV)
    mask_p = bench.circle_mask(P).expand(1, 3, P, P).contiguous().to(DEV)

    def make(n, B, seed):
        step = PatchAttackStep(n, args, B, H, W, device=DEV, shared_patch=True, patch_hw=(P, P))
        tgt, ref, origins = bench.synthetic_batch(B, seed, DEV)
        target = torch.randn(B, 2, H, W, generator=g).to(DEV)
        step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
        step.run(0)
        step.enqueue(4)
        torch.cuda.synchronize()
        return step

    full = make(net, 8, 1000)
    ha, hb = make(net, 4, 1001), make(net2, 4, 1002)
    N = 40

    def timed(fn):
        fn(4)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(N)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / N

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def one_stream(n):
        for _ in range(n):
            ha.enqueue(1)
            hb.enqueue(1)

    def two_streams(n):
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for _ in range(n):
            with torch.cuda.stream(s1):
                ha.enqueue(1)
            with torch.cuda.stream(s2):
                hb.enqueue(1)
        cur.wait_stream(s1); cur.wait_stream(s2)

    r = dict(ms_8_pairs_one_step=round(timed(full.enqueue), 3), ms_2x4_pairs_one_stream=round(timed(one_stream), 3),
             ms_2x4_pairs_two_streams=round(timed(two_streams), 3))
    r["ms_8_pairs_one_step_again"] = round(timed(full.enqueue), 3)
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

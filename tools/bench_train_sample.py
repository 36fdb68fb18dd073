"""One loader item of patch_attacks/main.py::train (clean forward, placement, attack of max_count=2
iterations, crop + zoom back) at 384x1280, batch 1: host placement (numpy/scipy, like the reference)
against the on-device placement.   python tools/bench_train_sample.py [samples]"""
import os
import sys
import time
from argparse import Namespace

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(samples=10):
    from understanding_flow_robustness_amd import utils_patch as up
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import train_sample, train_sample_device
    dev = "cuda:0"
    H, W = 384, 1280
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1.0e3, max_count=2, patch_type="circle")
    torch.backends.cudnn.benchmark = True
    net = fetch_model(args, synthetic_seed=0).to(dev)
    g = torch.Generator().manual_seed(0)
    frames = [(torch.rand(1, 3, H, W, generator=g).to(dev), torch.rand(1, 3, H, W, generator=g).to(dev)) for _ in range(4)]
    np.random.seed(1)
    p0, m0, sh0 = up.init_patch_circle(H, 0.1329)
    for name in ("host", "device"):
        np.random.seed(2)
        if name == "host":
            state = (p0.copy(), m0.copy(), p0.copy(), sh0)
            fn = train_sample
        else:
            f64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
            state = (f64(p0), f64(m0), f64(p0), sh0)
            fn = train_sample_device
        for k in range(samples + 3):
            if k == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            tgt, ref = frames[k % 4]
            state = fn(net, tgt, ref, ref, state[0], state[1], state[2], state[3], sh0, args)
        torch.cuda.synchronize()
        print(f"{name:6s} placement: {(time.perf_counter() - t0) / samples * 1e3:7.2f} ms per train() sample", flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10)

"""What plane WRITES get on this MI355X (round 6): streaming fills / copies of the sizes the igemm's epilogue and conv1_direct write,
timed with HIP events over back-to-back launches.  One JSON line per case."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters=30):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


for mb in (50, 94, 377, 1024):
    n = mb * 1000 * 1000 // 4
    x = torch.empty(n, dtype=torch.float32, device=DEV)
    y = torch.empty(n, dtype=torch.float32, device=DEV)
    ms = timed(lambda: x.fill_(1.0))
    print(json.dumps(dict(case=f"fill {mb} MB (write only)", ms=round(ms, 4), write_TBs=round(mb / ms / 1e3, 2))), flush=True)
    ms = timed(lambda: y.copy_(x))
    print(json.dumps(dict(case=f"copy {mb} MB (read + write)", ms=round(ms, 4), write_TBs=round(mb / ms / 1e3, 2), total_TBs=round(2 * mb / ms / 1e3, 2))), flush=True)

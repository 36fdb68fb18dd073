"""Per-kernel timing of the gfx950 correlation at BASELINE shapes (HIP events on the launch stream).

    python tools/microbench.py            # default kernels
    python tools/microbench.py sweep      # re-runs itself per UFR_CORR_{FWD,BWD}_VARIANT value
Each run also checks the selected variant against the general path (variant 4) on the same inputs.
"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters=20, warm=3):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def run_once():
    import torch
    from understanding_flow_robustness_amd import spatial_correlation_sampler_backend as be
    dev = "cuda:0"
    res = {}
    g = torch.Generator().manual_seed(0)
    for B in (1, 8):
        a = torch.randn(B, 256, 48, 160, generator=g).to(dev)
        b = torch.randn(B, 256, 48, 160, generator=g).to(dev)
        prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
        out = be.forward(a, b, *prm)
        go = torch.randn(out.shape, generator=g).to(dev)
        g1, g2 = be.backward(a, b, go, *prm)
        tf = timeit(lambda: be.forward(a, b, *prm))
        tb = timeit(lambda: be.backward(a, b, go, *prm))
        res[f"flownetc_B{B}"] = dict(fwd_ms=round(tf, 4), bwd_ms=round(tb, 4), fwd_tflops=round(1.734 * B / tf, 2),
                                     bwd_tflops=round(3.468 * B / tb, 2),
                                     chk=[float(out.double().abs().sum()), float(g1.double().abs().sum()),
                                          float(g2.double().abs().sum())])
    for (C, H, W) in ((196, 6, 20), (128, 12, 40), (96, 24, 80), (64, 48, 160), (32, 96, 320)):
        B = 8
        a = torch.randn(B, C, H, W, generator=g).to(dev)
        b = torch.randn(B, C, H, W, generator=g).to(dev)
        prm = (1, 1, 9, 9, 0, 0, 1, 1, 1, 1, 1, 1)
        out = be.forward(a, b, *prm)
        go = torch.randn(out.shape, generator=g).to(dev)
        g1, g2 = be.backward(a, b, go, *prm)
        res[f"pwc_C{C}_{H}x{W}_B8"] = dict(fwd_ms=round(timeit(lambda: be.forward(a, b, *prm)), 4),
                                          bwd_ms=round(timeit(lambda: be.backward(a, b, go, *prm)), 4),
                                          chk=[float(out.double().abs().sum()), float(g1.double().abs().sum()),
                                               float(g2.double().abs().sum())])
    print(json.dumps(res))


def sweep():
    base = None
    combos = [(4, 4, 0, 0), (1, 0, 0, 0), (1, 0, 1, 1), (0, 1, 1, 1), (2, 2, 1, 1), (3, 0, 1, 0)]
    if len(sys.argv) > 2:
        combos = [combos[0]] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[2:]]
    for fv, bv, fs, bs in combos:
        env = dict(os.environ, UFR_CORR_FWD_VARIANT=str(fv), UFR_CORR_BWD_VARIANT=str(bv),
                   UFR_CORR_FWD_SWZ=str(fs), UFR_CORR_BWD_SWZ=str(bs))
        out = subprocess.run([sys.executable, __file__], env=env, capture_output=True, text=True)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if not line:
            print(f"variant fwd={fv} bwd={bv}: FAILED\n{out.stderr[-2000:]}")
            continue
        res = json.loads(line[-1])
        if base is None:
            base = res
        ok = all(abs(x - y) <= 1e-5 * abs(y) for k in res for x, y in zip(res[k]["chk"], base[k]["chk"]))
        print(f"== fwd variant {fv} (XCD swizzle {fs}), bwd variant {bv} (XCD swizzle {bs}): checksums "
              f"{'match' if ok else 'DIFFER from'} the general path")
        for k, v in res.items():
            print(f"   {k:24s} fwd {v['fwd_ms']:8.4f} ms  bwd {v['bwd_ms']:8.4f} ms"
                  + (f"   ({v['fwd_tflops']} / {v['bwd_tflops']} TFLOP/s)" if "fwd_tflops" in v else ""))


if __name__ == "__main__":
    sweep() if len(sys.argv) > 1 and sys.argv[1] == "sweep" else run_once()

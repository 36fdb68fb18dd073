"""The clock the igemm really runs at (measured, not inferred).

    python tools/measure_clock.py [--seconds 2.5] [--layers conv3_1,conv4_1]

Builds the headline step (FlowNetC 384x1280, 8 pairs), runs two iterations so that every buffer holds real activations, then for
each chosen launch of the ping-pong kernel (`igemm_pp_kernel`, csrc/igemm.hip):
  1. replays it back to back for >= --seconds (a sustained load) while a host thread samples the SMI's sclk;
  2. in the LAST launch every workgroup records s_memtime (core cycles) and s_memrealtime (constant 100 MHz) at entry and exit
     (`ufr_igemm_clock_probe`): cycles / real time = MHz per workgroup.
Prints one JSON object per launch: the per-workgroup clock (median, 5th / 95th percentile), the SMI samples, the launch's
duration from HIP events, and what fraction of the six-product ceiling it delivers at the nominal 2.4 GHz and at the MEASURED
clock.  The same for a cold start (one launch after 50 ms of idle) shows the boost clock the kernel starts from.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import threading
import time
from argparse import Namespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from understanding_flow_robustness_amd import _lib as L  # noqa: E402

DEV = "cuda:0"
NOMINAL_MHZ = 2400.0
PEAK_SPLIT6_TFLOPS = 2500.0 / 6


def smi_sclk():
    """Current shader clock in MHz from rocm-smi (sysfs under the hood; no HIP call), or None."""
    for cmd in (["rocm-smi", "--showclocks"], ["amd-smi", "metric", "--clock"]):
        try:
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=10).stdout
        except (OSError, subprocess.TimeoutExpired):
            continue
        m = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)\s*Mhz", out, re.I) or re.search(r"GFX_0:\s*\n\s*CLK:\s*(\d+)\s*MHz", out)
        if m:
            return int(m.group(1))
    return None


def pm_info_sclk():
    """sysfs fallback: the starred line of pp_dpm_sclk of the first amdgpu card."""
    import glob
    for f in sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")):
        try:
            for ln in open(f):
                if "*" in ln:
                    return int(re.search(r"(\d+)\s*Mhz", ln, re.I).group(1))
        except (OSError, AttributeError):
            continue
    return None


class Sampler(threading.Thread):
    def __init__(self, period=0.25):
        super().__init__(daemon=True)
        self.period, self.samples, self.stop = period, [], False

    def run(self):
        while not self.stop:
            v = pm_info_sclk() or smi_sclk()
            if v:
                self.samples.append(v)
            time.sleep(self.period)


def probe_launch(launch, seconds):
    d = launch.desc
    n_wg = (d.Npad // 128) * ((d.B * d.Hr * d.Wr + 255) // 256) * 64      # upper bound on z (phases x split-K slices)
    buf = torch.zeros(n_wg, 8, dtype=torch.int64, device=DEV)
    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); launch(); e.record(); e.synchronize()
    one = s.elapsed_time(e)
    reps = max(10, int(seconds * 1e3 / max(one, 1e-3)))
    sampler = Sampler()
    sampler.start()
    s.record()
    for _ in range(reps):
        launch()
    e.record()
    e.synchronize()
    sustained_ms = s.elapsed_time(e) / reps
    # the probed launch rides directly behind the sustained run (the probe switch itself is a synchronous 12-byte copy)
    L.check(L.lib().ufr_igemm_clock_probe(L.ptr(buf), n_wg), "clock probe on")
    for _ in range(20):
        launch()
    torch.cuda.synchronize()
    sampler.stop = True
    L.check(L.lib().ufr_igemm_clock_probe(None, 0), "clock probe off")
    hot = buf.cpu().clone()
    # cold start: idle, then ONE probed launch
    time.sleep(0.2)
    buf.zero_()
    L.check(L.lib().ufr_igemm_clock_probe(L.ptr(buf), n_wg), "clock probe on")
    launch()
    torch.cuda.synchronize()
    L.check(L.lib().ufr_igemm_clock_probe(None, 0), "clock probe off")
    cold = buf.cpu().clone()

    def mhz(t):
        t = t[t[:, 3] > t[:, 2]]
        cyc, real = (t[:, 1] - t[:, 0]).double(), (t[:, 3] - t[:, 2]).double()
        f = (cyc / real * 100.0).sort().values                  # s_memrealtime ticks at 100 MHz
        q = lambda p: round(float(f[min(len(f) - 1, int(p * len(f)))]), 1)
        med = lambda v: int(v.double().median())
        t0 = int(t[:, 2].min())
        starts = ((t[:, 2] - t0).double() / 100.0).sort().values          # us after the launch's first workgroup
        return dict(workgroups=len(f), median=q(0.5), p05=q(0.05), p95=q(0.95),
                    wg_cycles_median=int(cyc.median()), wg_us_median=round(float(real.median()) / 100.0, 2),
                    k_steps=med(t[:, 7]), setup_cycles=med(t[:, 4] - t[:, 0]), loop_cycles=med(t[:, 5] - t[:, 4]),
                    epilogue_cycles=med(t[:, 1] - t[:, 5]), epilogue_p05_p95=[int((t[:, 1] - t[:, 5]).double().quantile(p)) for p in (0.05, 0.95)],
                    launch_span_us=round(float((t[:, 3].max() - t0)) / 100.0, 1),
                    start_us_quantiles=[round(float(starts[min(len(starts) - 1, int(p * len(starts)))]), 1) for p in (0.25, 0.5, 0.6, 0.75, 0.95)])
    return dict(reps=reps, ms_single=round(one, 4), ms_sustained=round(sustained_ms, 4), clock_hot=mhz(hot), clock_cold=mhz(cold),
                smi_sclk_samples=sampler.samples[:40])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--layers", default="conv3_1,conv4_1")
    ap.add_argument("--tags", default="full", help="full | band | window | prefix (comma separated): which of the engine's launch tables")
    opt = ap.parse_args()
    import bench
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    args = Namespace(flownet="FlowNetC", l2=False, alpha=0.0, lr=1000.0, max_count=2)
    net = fetch_model(args, synthetic_seed=0).to(DEV)
    B, H, W, P = bench.B_PER_GPU, bench.H, bench.W, bench.PATCH
    step = PatchAttackStep(net, args, B, H, W, device=DEV, shared_patch=True, patch_hw=(P, P))
    tgt, ref, origins = bench.synthetic_batch(B, 1000, DEV)
    with torch.no_grad():
        target = -torch.cat([net(tgt[i:i + 1], ref[i:i + 1]) for i in range(B)])
    g = torch.Generator().manual_seed(7)
    patch0 = torch.rand(1, 3, P, P, generator=g).to(DEV)
    mask_p = bench.circle_mask(P).expand(1, 3, P, P).contiguous().to(DEV)
    step.load(tgt, ref, patch0, mask_p, patch0, target, origins=origins)
    step.run(2)                                                    # real activations and gradients in every buffer
    eng = step.eng
    wanted = set(opt.layers.replace("+", ",").split(","))
    print(json.dumps(dict(sclk_idle=pm_info_sclk() or smi_sclk(), nominal_mhz=NOMINAL_MHZ)), flush=True)
    for name, kind, tag, launch, gflop in eng.launch_table():
        if name not in wanted or tag not in opt.tags.replace("+", ",").split(",") or launch.desc.variant != 6:
            continue
        r = probe_launch(launch, opt.seconds)
        mhz = r["clock_hot"]["median"]
        tf = gflop / r["ms_sustained"]
        r.update(launch=f"{name} {kind} {tag}", splitk=launch.desc.splitk, gflop=round(gflop, 2), tflops_sustained=round(tf, 1),
                 frac_of_ceiling_at_nominal_clock=round(tf / PEAK_SPLIT6_TFLOPS, 3),
                 frac_of_ceiling_at_measured_clock=round(tf / (PEAK_SPLIT6_TFLOPS * mhz / NOMINAL_MHZ), 3))
        print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

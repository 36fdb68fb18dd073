"""bench.py -- headline metric of BASELINE.json on MI355X.

    python bench.py [--gpus N --steps K --warmup W] [--config c2|c4|c5]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

metric   : attack-iters/s = frame pairs x I-FGSM patch iterations per second
workload : configs[1] -- FlowNetC, 384x1280 synthetic frame pairs, batch 8 per GPU, ONE 51x51 circular patch in
           patch coordinates shown by every pair at its own placement, cosine loss, lr 1000
           (patch_attacks/main.py defaults), fp32 end to end.
step     : ONE inner-loop iteration of attack() (main.py:546-611) over the rank's batch:
           paste -> FlowNetC forward -> loss -> data-gradient backward -> crop to [3,51,51] + sum -> clamp/update/re-paste,
           replayed as one HIP graph (N>1: two graphs around one RCCL all-gather of the 31 KB [crop | loss] rows).
scaling  : weak (8 pairs per GPU); value = N*8*K / max-over-ranks time.
Inputs are resident in HBM before the timed region.  The JSON line also carries
  roofline     -- the dominant kernel (csrc/igemm.hip: the head's convolutions, float32-accurate on the bf16 matrix cores
                  with six products -> ceiling 2.5 PFLOP/s / 6 = 417 TFLOP/s fp32-equivalent) over the launches of one
                  iteration, each timed live with HIP events; under "kernels" every launch of the step's own kernels and
                  the MIOpen prefix against its own bound; under "step" the whole iteration's executed FLOPs;
  cpu_baseline -- the CPU oracle (torch-CPU convs + C correlation) on a bounded sample, rank 0, N=1.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from argparse import Namespace

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W, B_PER_GPU = 384, 1280, 8
PATCH = 51                      # --patch-size 0.1329 * 384 (utils_patch.py:760-766)
# algorithmic work per frame pair per iteration, forward + data gradient (SURVEY.md 8d / BASELINE.md)
GFLOP_NET_FWD = 111.9           # FlowNetC convolutions, one pair, forward
GFLOP_PREFIX_FWD = 2 * (2.312 + 12.583 + 12.583)   # conv1-3 on both frames (2*Cin*Cout*k*k*Hout*Wout)
GFLOP_CORR_FWD, GFLOP_CORR_BWD = 1.734, 3.468
GFLOP_CORR = GFLOP_CORR_FWD + GFLOP_CORR_BWD
GFLOP_CORR_BWD_WINDOW = 2 * 2 * 256 * 441 * 16 * 16 / 1e9       # both adjoints on the window's 16x16 cells (0.116 per pair)


GFLOP_BAND_LAYERS = 16.74 + 4.53 + 9.06 + 2.26     # data gradients of conv3_1, conv4, conv4_1, conv5 (one pair)


GFLOP_INC_LAYERS = 16.74 + 4.53 + 9.06            # forward of conv3_1, conv4, conv4_1 (one pair)


def gflop_per_pair_step(window_hw, max_count, band_width=None, incremental=False):
    """FLOPs the step EXECUTES per pair and iteration.  Full-frame: whole network forward + data gradient.
    Windowed prefix (patch_attack.py): head forward + adjoint at full size, conv1-3 forward + adjoint on
    the window, the correlation's adjoint on the window's cells, plus the one full-frame conv1-3 forward per attack()
    call spread over its iterations; with a column band four head data gradients shrink to band_width / W."""
    if window_hw is None:
        return 2 * GFLOP_NET_FWD + GFLOP_CORR
    frac = window_hw[0] * window_hw[1] / float(H * W)
    head = GFLOP_NET_FWD - GFLOP_PREFIX_FWD
    total = 2 * head + GFLOP_CORR_FWD + GFLOP_CORR_BWD_WINDOW + 2 * GFLOP_PREFIX_FWD * frac + GFLOP_PREFIX_FWD / max_count
    if band_width is not None:
        total -= GFLOP_BAND_LAYERS * (1.0 - band_width / float(W))
        if incremental:    # conv3_1 / conv4 / conv4_1 forward on the band only, from the 2nd iteration of a call on
            total -= GFLOP_INC_LAYERS * (1.0 - band_width / float(W)) * (max_count - 1) / max_count
    return total


GFLOP_PER_PAIR_STEP = gflop_per_pair_step(None, 2)
PEAK_FP32_TFLOPS = 157.3        # MI355X_MICROARCH.md: fp32 vector == fp32 MFMA peak
PEAK_SPLIT6_TFLOPS = 2500.0 / 6   # bf16 dense MFMA peak / six products per float32 product (csrc/igemm.hip)
PEAK_HBM_GBS = 8000.0
NOMINAL_MHZ = 2400.0             # the clock the 2.5 PFLOP/s ceiling is priced at
CORR_FWD = dict(gflop=1.734, mbytes=29.3)   # per [1,256,48,160] pair
CORR_BWD = dict(gflop=3.468, mbytes=44.9)


def circle_mask(size):
    yy, xx = torch.meshgrid(torch.arange(size), torch.arange(size), indexing="ij")
    c = size // 2
    return ((yy - c) ** 2 + (xx - c) ** 2 <= (c - 2) ** 2).float()   # utils_patch.py:236-247


def synthetic_batch(batch, seed, device):
    """Frames + one seeded random placement (row, column) per pair (circle_transform places randomly)."""
    g = torch.Generator().manual_seed(seed)
    tgt = torch.rand(batch, 3, H, W, generator=g)
    ref = torch.rand(batch, 3, H, W, generator=g)
    origins = [(int(torch.randint(0, H - PATCH, (1,), generator=g)), int(torch.randint(0, W - PATCH, (1,), generator=g)))
               for _ in range(batch)]
    return tgt.to(device), ref.to(device), origins


def event_time(fn, iters, warm=2):
    """Average duration (ms) of `fn` with HIP events on the stream the kernels are launched on."""
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / iters


# PMC summaries of the latest passes over this command (tools/gpu_call.sh `traffic` step -> tools/pmc_step_traffic.py)
IGEMM_TRAFFIC, STEP_TRAFFIC, CORR_TRAFFIC = "r6_igemm_traffic.json", "r6_step_traffic.json", "r4_corr_planes_traffic.json"


def _pmc(name):
    try:
        with open(os.path.join(ROOT, "profiles", name)) as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def kernel_rooflines(step, device, max_count):
    """Live timings (HIP events on the launch stream) of every kernel class of the step on the step's own shapes.
    Returns (entries, igemm aggregate over one average iteration)."""
    from understanding_flow_robustness_amd import _lib as L
    from understanding_flow_robustness_amd import spatial_correlation_sampler_backend as be
    B = B_PER_GPU
    ks = []
    traffic = _pmc(IGEMM_TRAFFIC)                     # HBM-side bytes per launch from the PMC passes (rocprofv3 only)
    # ---- the head's convolutions: every prepared igemm launch of the engine the step runs
    eng = None
    for e in getattr(step.net, "__dict__", {}).get("_ufr_head_engines", {}).values():
        if e.B == B and e.H == H and e.W == W:
            eng = e
    agg = None
    if eng is not None:
        per = {}
        for name, kind, tag, launch, gflop in eng.launch_table():
            t = event_time(launch, 10)
            tf = gflop / t
            key = f"igemm {name} {kind} ({tag})"
            per[(name, kind, tag)] = (t, gflop, launch.algorithmic_bytes())
            ks.append(dict(kernel=key, ms=round(t, 4), bound="mfma", achieved=round(tf, 1), peak=round(PEAK_SPLIT6_TFLOPS, 1),
                           unit="TFLOP/s", frac=round(tf / PEAK_SPLIT6_TFLOPS, 3), gflop=round(gflop, 2),
                           traffic=traffic.get(key)))
        # one average iteration of an attack() call of `max_count` iterations: the first runs the full forward, later
        # ones the band's columns of conv_redir / conv3_1 / conv4 / conv4_1; five data gradients always run on the band
        banded = bool(eng.bwd_band)
        t_sum = g_sum = b_sum = 0.0
        for (name, kind, tag), (t, gflop, nbytes) in per.items():
            if tag == "window":
                w = 1.0
            elif tag == "prefix":              # full-frame conv2 / conv3 of load(): once per attack() call
                w = 1.0 / max_count
            elif kind == "fwd":
                has_band = (name, "fwd", "band") in per
                w = (1.0 / max_count if has_band else 1.0) if tag == "full" else (max_count - 1.0) / max_count
            else:
                has_band = banded and (name, "bwd", "band") in per
                w = (0.0 if has_band else 1.0) if tag == "full" else 1.0
            t_sum += w * t
            g_sum += w * gflop
            b_sum += w * nbytes
        agg = dict(ms=t_sum, gflop=g_sum, bytes=b_sum)
    # ---- conv1 straight from the raw frames (csrc/conv1_direct.hip): HBM-bound -- frames in, three bf16 planes of conv1 out
    if eng is not None and hasattr(eng, "conv1_direct_table"):
        for label, tag, fn, nbytes, gflop in eng.conv1_direct_table():
            t = event_time(fn, 10)
            ks.append(dict(kernel=f"{label} ({tag})", ms=round(t, 4), bound="hbm", achieved=round(nbytes / t / 1e6, 1), peak=PEAK_HBM_GBS,
                           unit="GB/s", frac=round(nbytes / t / 1e6 / PEAK_HBM_GBS, 4), algorithmic_bytes=int(nbytes),
                           tflops=round(gflop / t, 1), traffic=None))
    # ---- correlation
    a = torch.randn(B, 256, H // 8, W // 8, device=device)
    b = torch.randn(B, 256, H // 8, W // 8, device=device)
    prm = (1, 1, 21, 21, 0, 0, 1, 1, 2, 2, 1, 1)
    out = be.forward(a, b, *prm)
    go = torch.randn_like(out)
    if eng is not None:
        # the step's cost volume: banded GEMM on the matrix cores, planes in / planes out (csrc/correlation_planes.hip).
        # Algorithmic bytes: both feature maps' planes in (3 x bf16) + the 441 channels' planes out; flops as fp32 MACs.
        from understanding_flow_robustness_amd import igemm as ig
        h8, w8 = H // 8, W // 8
        t_f = event_time(lambda: L.check(L.lib().ufr_corr_forward_planes(
            L.ptr(eng.c3a_p.t), L.ptr(eng.c3b_p.t), eng.c3a_p.plane_stride, L.ptr(eng.in31.t), eng.in31.plane_stride, 1, B, 256,
            h8, w8, 21, 2, 1.0 / 256.0, ig.LEAKY, L.stream())), 10)
        tf = CORR_FWD["gflop"] * B / t_f
        nbytes = B * h8 * w8 * (2 * 256 + 441) * 6
        ks.append(dict(kernel="corr_fwd_planes_k2_kernel<5> (21x21 cost volume, / C, LeakyReLU, planes in / out)", ms=round(t_f, 4), bound="mfma",
                       achieved=round(tf, 2), peak=PEAK_FP32_TFLOPS, unit="TFLOP/s", frac=round(tf / PEAK_FP32_TFLOPS, 4),
                       hbm_gbs=round(nbytes / t_f / 1e6, 1), traffic=_pmc(CORR_TRAFFIC).get("traffic_bytes"),
                       algorithmic_bytes=int(nbytes),
                       note="useful fp32 flops against the fp32 vector peak; on the matrix cores 44 % of the six-product MFMA work lies in the band"))
    else:
        t_f = event_time(lambda: be.forward(a, b, *prm), 10)
        pmc = _pmc("r1_corr_traffic.json")
        tf = CORR_FWD["gflop"] * B / t_f
        ks.append(dict(kernel="corr_fwd_vec<21,2>", ms=round(t_f, 4), bound="mfma", achieved=round(tf, 2), peak=PEAK_FP32_TFLOPS,
                       unit="TFLOP/s", frac=round(tf / PEAK_FP32_TFLOPS, 4), hbm_gbs=round(CORR_FWD["mbytes"] * B / t_f, 1),
                       traffic=(pmc["corr_fwd_vec<21,2>"]["fetch_bytes"] + pmc["corr_fwd_vec<21,2>"]["write_bytes"]) if "corr_fwd_vec<21,2>" in pmc else None,
                       algorithmic_bytes=int(CORR_FWD["mbytes"] * 1e6 * B)))
    win = torch.zeros(B, 8, dtype=torch.int32, device=device)
    win[:, 0], win[:, 1] = 128, 512
    cells = 16 * 16
    if eng is not None:
        # the step's form: both adjoints on the window's cells on the matrix cores, gradient sums in, window gradient out
        # (csrc/correlation_window_mfma.hip).  Algorithmic bytes per pair: the band of cost-volume gradients of the window's
        # cells (both adjoints), the two 56 x 64-cell feature regions, conv_redir's window gradient, the window-sized output.
        gw = torch.empty(2 * B, 256, 16, 16, device=device)
        t_w = event_time(lambda: L.check(L.lib().ufr_corr_backward_window_fused(
            L.ptr(a), L.ptr(b), L.ptr(eng.G_in31.t), 1, 1.0 / 256.0, L.ptr(eng.G_c3a.t), L.ptr(gw), B, 256, H // 8, W // 8, 21, 2,
            L.ptr(win), 8, 16, 16, 1, L.stream())), 10)
        bytes_w = B * (2 * 441 * cells * 4 + 2 * 256 * 56 * 64 * 4 + 256 * cells * 4 + 2 * 256 * cells * 4)
        name = "corr_bwd_window_mfma_kernel<2,2> (both adjoints + conv_redir's gradient, 16x16 cells per pair, window-sized output)"
    else:
        g1, g2 = torch.empty_like(a), torch.empty_like(b)
        t_w = event_time(lambda: L.check(L.lib().ufr_corr_backward_window(
            L.ptr(a), L.ptr(b), L.ptr(go), L.ptr(g1), L.ptr(g2), B, 256, H // 8, W // 8, 21, 2, L.ptr(win), 8, 16, 16,
            L.stream())), 10)
        bytes_w = B * (2 * 441 * cells * 4 + 2 * 256 * 56 * 56 * 4 + 2 * 256 * (H // 8) * (W // 8) * 4)   # gout, regions, zero-filled outputs
        name = "corr_bwd_window<8,21,2> (both adjoints, 16x16 cells per pair)"
    ks.append(dict(kernel=name, ms=round(t_w, 4), bound="hbm",
                   achieved=round(bytes_w / t_w / 1e6, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                   frac=round(bytes_w / t_w / 1e6 / PEAK_HBM_GBS, 4), tflops=round(GFLOP_CORR_BWD_WINDOW * B / t_w, 2),
                   traffic=_pmc("r5_corr_window_traffic.json").get("traffic_bytes"), algorithmic_bytes=int(bytes_w)))
    # ---- the 2-channel layers of the refinement (HBM-bound: one pass over the concatenation's planes)
    if eng is not None:
        for k in (6, 5, 4, 3, 2):
            src, chunks = eng.pf_src[k]
            nbytes = chunks * src.M * 32 * 6 + B * 2 * src.H * src.W * 4
            t = event_time(lambda: eng._pf_forward(k), 10)
            ks.append(dict(kernel=f"flow_head_planes_fwd predict_flow{k}", ms=round(t, 4), bound="hbm", achieved=round(nbytes / t / 1e6, 1),
                           peak=PEAK_HBM_GBS, unit="GB/s", frac=round(nbytes / t / 1e6 / PEAK_HBM_GBS, 4), algorithmic_bytes=int(nbytes), traffic=None))
            gsrc = eng.g_flow[k] if k != 2 else torch.zeros_like(eng.flow[2])
            nbytes_b = chunks * src.M * 32 * 4 * (1 if k in (2, 6) else 2) + B * 2 * src.H * src.W * 4
            t = event_time(lambda: eng._pf_backward(k, gsrc, accumulate=k not in (2, 6)), 10)
            ks.append(dict(kernel=f"flow_head_planes_bwd predict_flow{k}", ms=round(t, 4), bound="hbm", achieved=round(nbytes_b / t / 1e6, 1),
                           peak=PEAK_HBM_GBS, unit="GB/s", frac=round(nbytes_b / t / 1e6 / PEAK_HBM_GBS, 4), algorithmic_bytes=int(nbytes_b), traffic=None))
    return ks, agg


def step_traffic():
    """HBM-side bytes of one iteration of the windowed step from this round's PMC passes
    (tools/pmc_step_traffic.py -> profiles/r4_step_traffic.json); counters cannot be read live."""
    try:
        return int(_pmc(STEP_TRAFFIC)["traffic_bytes_per_iteration"])
    except (KeyError, ValueError):
        return None


def measured_clock(eng, device, seconds=0.4):
    """The shader clock the dominant kernel really runs at, measured in THIS process: conv3_1's forward launch
    (ping-pong igemm, csrc/igemm.hip) replayed back to back for `seconds`, then one probed launch in which every workgroup records
    s_memtime (core cycles) and s_memrealtime (constant 100 MHz) at entry and exit (`ufr_igemm_clock_probe`): cycles / real time.
    Returns (median MHz over the workgroups, that launch's sustained TFLOP/s) or (None, None)."""
    from understanding_flow_robustness_amd import _lib as L
    row = next((r for r in eng.launch_table() if r[0] == "conv3_1" and r[1] == "fwd" and r[2] == "full" and r[3].desc.variant == 6), None)
    if row is None:
        return None, None
    launch, gflop = row[3], row[4]
    d = launch.desc
    n_wg = (d.Npad // 128) * ((d.B * d.Hr * d.Wr + 255) // 256) * 64
    buf = torch.zeros(n_wg, 8, dtype=torch.int64, device=device)
    one = event_time(launch, 5)
    reps = max(10, int(seconds * 1e3 / max(one, 1e-3)))
    sustained = event_time(launch, reps, warm=0)
    L.check(L.lib().ufr_igemm_clock_probe(L.ptr(buf), n_wg), "clock probe on")
    try:
        for _ in range(4):
            launch()
        torch.cuda.synchronize(device)
    finally:
        L.check(L.lib().ufr_igemm_clock_probe(None, 0), "clock probe off")
    t = buf.cpu()
    t = t[t[:, 3] > t[:, 2]]
    if not len(t):
        return None, None
    mhz = ((t[:, 1] - t[:, 0]).double() / (t[:, 3] - t[:, 2]).double() * 100.0).median()
    return round(float(mhz), 1), round(gflop / sustained, 1)


def cpu_baseline():
    """The CPU oracle on a bounded sample of the same workload: ONE attack() call of 2 iterations on
    one 384x1280 pair (the reference's only batch size), all host threads."""
    from oracle import flow_oracle as fo
    from oracle import oracle_ops
    from understanding_flow_robustness_amd.flownets.flownetc import FlowNetC
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    oracle_ops.lib()
    # the box's share for one GPU is 16 host cores (256 are visible; oversubscribing them is slower)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    oracle_ops.lib().ufr_oracle_set_threads(cores)
    sd = synthetic_state_dict(FlowNetC().state_dict(), seed=0)
    tgt, ref, origins = synthetic_batch(1, 1234, "cpu")
    mask = fo.place(circle_mask(PATCH).expand(1, 3, PATCH, PATCH), origins, H, W)      # the reference's canvas form
    g = torch.Generator().manual_seed(99)
    patch0 = torch.rand(1, 3, H, W, generator=g) * mask
    predict = lambda x, y: fo.flownetc_forward(sd, x, y)
    with torch.no_grad():
        target = -predict(tgt, ref)           # clean forward doubles as the warm-up (main.py:371,395)
        t1 = time.time()
        predict(tgt, ref)                     # configs[0] (C1): FlowNetC forward on one 384x1280 pair, CPU only
        c1 = time.time() - t1
    t0, n, calls = time.time(), 0, 0
    while time.time() - t0 < 12.0:            # bounded sample: ~12 s of CPU work
        _, _, _, k, _ = fo.patch_attack(predict, tgt, ref, patch0.clone(), mask, patch0, target, lr=1e3, max_count=2)
        n, calls = n + k, calls + 1
    dt = time.time() - t0
    return dict(value=round(n / dt, 4), unit="frame-pairs*steps/s", cores=cores, cpu_model=cpu_model(), kind="port",
                c1_cpu_forward_s=round(c1, 3),
                sample=f"{calls} attack() calls of 2 iterations ({n} iterations), 1 pair 384x1280 (the reference's "
                       f"batch size), FlowNetC fp32, torch-CPU convs + C oracle correlation (OpenMP over "
                       f"batch*channels; the reference's CPU backward is single-threaded at batch 1), {dt:.1f} s")


def launch_ranks(n, argv):
    """One process per GPU through `python -m torch.distributed.run --standalone` on 127.0.0.1 (the launcher picks and owns
    its rendezvous port: no bind-then-close race), as a CHILD process: this process has made no HIP call yet and makes none.
    Returns the exit code; rank 0's stdout (the one JSON line) is relayed, everything else goes to stderr on failure."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1",
           f"--nproc-per-node={n}", os.path.abspath(__file__)] + list(argv)
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    for ln in lines[-1:]:
        print(ln, flush=True)
    if proc.returncode != 0:                                # a rank's diagnostic prints must not be lost
        sys.stderr.write("".join(ln + "\n" for ln in proc.stdout.splitlines() if ln not in lines[-1:]))
    if proc.returncode == 0 and not lines:
        print("bench.py: the ranks exited without a result line", file=sys.stderr)
        return 1
    return proc.returncode


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


# ------------------------------------------------------------------------------------ the three shardable configs
# Every config is (step, resident batches, how one attack() call is loaded, iterations per call): SURVEY.md 8e shards the
# pairs of C2 / C4 eight per rank behind ONE patch (31 KB all-gather per iteration) and the pairs of C5 one per rank behind
# ONE perturbation (11 MiB all-reduce per iteration).
CONFIG_WORKLOADS = {
    "c2": "FlowNetC 384x1280 I-FGSM patch attack (configs[1])",
    "c4": "PWC-Net 384x1280 I-FGSM patch attack (configs[3]: batch 64 sharded 8 per GPU, patch-gradient exchange)",
    "c5": "FlowNet2 448x1024 universal-perturbation loop (configs[4]: one pair per GPU, image-sized all-reduce)",
}


def setup_patch_config(opt, rank, world, device, flownet, seed):
    """C2 / C4: `B_PER_GPU` pairs per rank behind one 51x51 patch in patch coordinates (patch_attacks/main.py:523-613)."""
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep, ShardedExchange
    args = Namespace(flownet=flownet, l2=False, alpha=0.0, lr=1000.0, max_count=2)
    net = fetch_model(args, synthetic_seed=seed).to(device)
    exchange = ShardedExchange() if world > 1 else None
    step = PatchAttackStep(net, args, B_PER_GPU, H, W, device=device, shared_patch=True, exchange=exchange,
                           use_graph=not opt.no_graph, warmup=2, patch_hw=(PATCH, PATCH))
    # two resident batches (frames AND patch placements differ): consecutive attack() calls never see
    # the same operands, so nothing cached for one call can serve the next
    g = torch.Generator().manual_seed(7)
    patch0 = torch.rand(1, 3, PATCH, PATCH, generator=g).to(device)     # same patch on every rank
    mask_p = circle_mask(PATCH).expand(1, 3, PATCH, PATCH).contiguous().to(device)
    batches = []
    for k in range(2):
        tgt, ref, origins = synthetic_batch(B_PER_GPU, 1000 + 17 * k + rank, device)
        with torch.no_grad():
            target = -torch.cat([predict_flow(net, None, tgt[i:i + 1], ref[i:i + 1], args) for i in range(B_PER_GPU)])   # main.py:395
        batches.append(dict(args=(tgt, ref, patch0, mask_p, patch0, target), origins=origins))
    load = lambda call: step.load(*batches[call % 2]["args"], origins=batches[call % 2]["origins"])
    load(0)
    step.run(0)                                                  # warm-up + graph capture, reloads operands
    return dict(net=net, args=args, step=step, batches=batches, load=load, pairs=B_PER_GPU, per_call=max(1, opt.max_count),
                executed=lambda: step.state[1], overflow=lambda: float(step.state[3]))


def setup_universal_config(opt, rank, world, device):
    """C5: one 448x1024 pair per rank behind ONE perturbation [2,3,H,W]; an attack() call = n_step = 10 sign steps
    (global_attacks/universal_perturbation.py:452-530, defaults :73-106), new frames per call."""
    from understanding_flow_robustness_amd.flownets.utils_model import fetch_model, predict_flow
    from understanding_flow_robustness_amd.patch_attack import ShardedExchange
    from understanding_flow_robustness_amd.universal_perturbation import UniversalPerturbationStep
    h, w = 448, 1024
    args = Namespace(flownet="FlowNet2", n_step=10, learning_rate=2e-3, output_norm=0.02, flow_loss="cossim",
                     perturb_method="ifgsm", perturb_mode="both", add_gaussian=False)
    net = fetch_model(args, synthetic_seed=3).to(device)
    exchange = ShardedExchange() if world > 1 else None
    step = UniversalPerturbationStep(net, args, 1, h, w, device=device, shared=True, exchange=exchange,
                                     use_graph=not opt.no_graph)
    batches = []
    for k in range(2):
        g = torch.Generator().manual_seed(2000 + 17 * k + rank)
        i0, i1 = torch.rand(1, 3, h, w, generator=g).to(device), torch.rand(1, 3, h, w, generator=g).to(device)
        with torch.no_grad():
            target = -predict_flow(net, None, i0, i1, args)
        batches.append((i0, i1, target))
    # the perturbation is carried from call to call like the reference's (`universal_perturbation_var`, :573-600)
    load = lambda call: step.load(*batches[call % 2][:2], step.delta, batches[call % 2][2])
    load(0)
    step.run(0)
    done = torch.zeros(1, device=device)       # no early exit in this loop (:464: `for _ in range(n_step)`): every enqueued step runs
    return dict(net=net, args=args, step=step, batches=batches, load=load, pairs=1, per_call=args.n_step, hw=(h, w),
                executed=None, overflow=lambda: 0.0, done=done)


def cpu_baseline_other(config):
    """C4 / C5 on the host cores: the CPU oracle (oracle/flow_oracle.py) on one pair, bounded to ~15 s."""
    from oracle import flow_oracle as fo
    from oracle import oracle_ops
    from understanding_flow_robustness_amd.flownets.weights import synthetic_state_dict
    oracle_ops.lib()
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))
    torch.set_num_threads(cores)
    oracle_ops.lib().ufr_oracle_set_threads(cores)
    t0, n = time.time(), 0
    if config == "c4":
        from understanding_flow_robustness_amd.flownets.pwcnet import PWCDCNet
        sd = synthetic_state_dict(PWCDCNet().state_dict(), seed=1)
        tgt, ref, origins = synthetic_batch(1, 1234, "cpu")
        mask = fo.place(circle_mask(PATCH).expand(1, 3, PATCH, PATCH), origins, H, W)
        patch0 = torch.rand(1, 3, H, W, generator=torch.Generator().manual_seed(99)) * mask
        predict = lambda x, y: fo.pwcnet_forward(sd, x, y)
        with torch.no_grad():
            target = -predict(tgt, ref)
        t0 = time.time()
        while time.time() - t0 < 12.0:
            n += fo.patch_attack(predict, tgt, ref, patch0.clone(), mask, patch0, target, lr=1e3, max_count=2)[3]
        what = "attack() calls of 2 iterations, 1 pair 384x1280, PWC-Net fp32"
    else:
        from understanding_flow_robustness_amd.flownets.flownet2 import FlowNet2
        sd = synthetic_state_dict(FlowNet2().state_dict(), seed=3)
        g = torch.Generator().manual_seed(1234)
        i0, i1 = torch.rand(1, 3, 448, 1024, generator=g), torch.rand(1, 3, 448, 1024, generator=g)
        predict = lambda x, y: fo.flownet2_forward(sd, x, y)
        with torch.no_grad():
            target = -predict(i0, i1)
        delta = torch.zeros(1, 2, 3, 448, 1024)
        t0 = time.time()
        while time.time() - t0 < 12.0:
            _, _, delta = fo.universal_attack(predict, i0, i1, delta, target, n_step=1, lr=2e-3, eps=0.02, flow_loss="cossim")
            n += 1
        what = "universal-perturbation sign steps, 1 pair 448x1024, FlowNet2 fp32"
    dt = time.time() - t0
    return dict(value=round(n / dt, 4), unit="frame-pairs*steps/s", cores=cores, cpu_model=cpu_model(), kind="port",
                sample=f"{n} {what}, torch-CPU convs + the C oracle's correlation / Resample2d / ChannelNorm, {dt:.1f} s")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIG_WORKLOADS), default="c2",
                    help="c2 = the headline (BASELINE configs[1]); c4 / c5 = the configs BASELINE shards over 8 GPUs")
    ap.add_argument("--sustained-seconds", type=float, default=2.0,
                    help="N=1: after the K timed steps, replay the same protocol for at least this long (0 = skip)")
    ap.add_argument("--precondition-seconds", type=float, default=0.0,
                    help="optional: before the W warm-up steps, this long of the same attack() calls, untimed (a launch after idle runs ~8 %% "
                         "slower clocked than in a sustained stream, profiles/r4_igemm_clock.txt; off by default: W warm-up steps, then K timed ones)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-full-frame", action="store_true", help="skip the full-frame side measurement")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the live side measurements of configs C3 / C4 / C5")
    ap.add_argument("--max-count", type=int, default=2, help="iterations per attack() call (main.py:79)")
    opt = ap.parse_args()

    if opt.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, BEFORE this process touches the GPU
        # (a child launcher, never a re-exec), relay rank 0's JSON line and fail if any rank fails
        raise SystemExit(launch_ranks(opt.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != opt.gpus:
        raise SystemExit(f"bench.py --gpus {opt.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    # one rank per GPU; on a box with fewer devices than ranks (the 1-GPU rehearsal of the N>1 path,
    # UFR_DIST_BACKEND=gloo) ranks share devices round-robin
    device = torch.device(f"cuda:{local % torch.cuda.device_count()}")
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("UFR_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    torch.backends.cudnn.benchmark = True      # patch_attacks/main.py:276 (MIOpen find mode)
    if opt.config == "c5":
        ctx = setup_universal_config(opt, rank, world, device)
    else:
        ctx = setup_patch_config(opt, rank, world, device, *(("FlowNetC", 0) if opt.config == "c2" else ("PWCNet", 1)))
    net, args, step, batches, pairs, mc = ctx["net"], ctx["args"], ctx["step"], ctx["batches"], ctx["pairs"], ctx["per_call"]
    executed_acc = torch.zeros(1, device=device)

    def attack_calls(iterations, first_call):
        """`iterations` inner-loop iterations as attack() calls of `per_call` (main.py:79 default 2; universal: n_step 10):
        every call loads a different batch (new frames, new placement: paste, window placement, full-frame prefix)
        and then replays the captured iteration.  Nothing is read back."""
        call, left = first_call, iterations
        while left > 0:
            ctx["load"](call)
            n = min(mc, left)
            step.enqueue(n)
            if ctx["executed"] is not None:
                executed_acc.add_(ctx["executed"]())
            else:
                executed_acc.add_(float(n))
            call, left = call + 1, left - n
        return call

    # the throughput run must never trip the early exit: an iteration that is skipped is not work done
    calls_done = 0
    if opt.precondition_seconds > 0:               # setup, like the graph capture: bring the chip to its steady clock
        t_pre = time.perf_counter()
        calls_done = attack_calls(4 * mc, calls_done)
        torch.cuda.synchronize(device)
        dt = max(time.perf_counter() - t_pre, 1e-4)
        more = torch.tensor([int(max(0.0, opt.precondition_seconds - dt) / dt)], device=device)
        if world > 1:                              # every rank must run the same number of exchanges
            dist.broadcast(more, src=0)
        for _ in range(int(more)):
            calls_done = attack_calls(4 * mc, calls_done)
        torch.cuda.synchronize(device)
    calls_done = attack_calls(opt.warmup, calls_done)
    torch.cuda.synchronize(device)
    executed0 = float(executed_acc)

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    calls_done = attack_calls(opt.steps, calls_done)
    torch.cuda.synchronize(device)
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    executed = float(executed_acc) - executed0
    if world > 1:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    if int(executed) != opt.steps:
        raise SystemExit(f"only {int(executed)} of {opt.steps} iterations took effect (loss gate tripped): "
                         "the timed region would contain skipped work")
    if ctx["overflow"]() != 0.0:
        raise SystemExit("a patch window overflowed inside the timed region")

    # a SUSTAINED figure beside the K-step one (N=1): the same calls for >= --sustained-seconds, so that a clock that sags
    # under a long load shows (the K = 20 region is ~0.1 s)
    sustained = None
    if world == 1 and opt.sustained_seconds > 0:
        ms_est = elapsed * 1e3 / opt.steps
        n_sus = int(-(-max(opt.sustained_seconds * 1e3 / ms_est, mc) // mc) * mc)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        attack_calls(n_sus, calls_done)
        torch.cuda.synchronize(device)
        dt = time.perf_counter() - t1
        sustained = dict(ms_per_step=round(dt * 1e3 / n_sus, 3), steps=n_sus, seconds=round(dt, 2))

    if rank == 0:
        ms = elapsed * 1e3 / opt.steps
        value = world * pairs * opt.steps / elapsed
        line = {
            "metric": "attack-iters/s", "value": round(value, 3), "unit": "frame-pairs*steps/s",
            "n_gpus": world, "steps": opt.steps, "warmup": opt.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": CONFIG_WORKLOADS[opt.config], "pairs_per_gpu": pairs, "global_pairs": world * pairs,
                       "preconditioning_s": opt.precondition_seconds,
                       "weights": "synthetic seeded (no checkpoints offline)", "graph": not opt.no_graph},
        }
        if sustained is not None:
            line["sustained_ms_per_step"] = sustained["ms_per_step"]
            line["config"]["sustained_ms_per_step"] = sustained["ms_per_step"]      # flat scalars: the driver's parser drops nested values
            line["config"]["sustained_steps"] = sustained["steps"]
            line["config"]["sustained"] = sustained
        if opt.config == "c2":
            c2_line(line, opt, ctx, world, device, ms, mc)
        else:
            other_line(line, opt, ctx, world, ms)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def other_line(line, opt, ctx, world, ms):
    """C4 / C5: the dominant kernel is the same igemm; its roofline = every prepared launch of the config's engines timed live
    on the step's own buffers (tools/bench_configs.igemm_roofline)."""
    cfg = line["config"]
    if opt.config == "c4":
        cfg.update(patch="ONE 51x51 circular patch in patch coordinates, shown by every pair at its own placement",
                   calls=f"attack() calls of max_count={ctx['per_call']} iterations, new frames + placement per call",
                   loss="cosine", lr=1000.0,
                   parallelism=f"dp{world}: pairs sharded, per-rank crop of the pre-clamp gradient to [3,51,51], all-gather of "
                               f"{world} x 31 KB rows + fixed-order sum before the clamp")
    else:
        cfg.update(calls="attack() calls of n_step=10 sign steps, new frames per call, the perturbation carried over",
                   loss="cossim", lr=2e-3, output_norm=0.02,
                   parallelism=f"dp{world}: one pair per rank, all-reduce(sum) of the [2,3,448,1024] gradient + loss (11 MiB) "
                               "before the sign")
    if world == 1:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs
        roof = bench_configs.igemm_roofline(ctx["net"])
        if roof is not None:
            line["roofline"] = roof
        if not opt.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline_other(opt.config)


def c2_line(line, opt, ctx, world, device, ms, mc):
    """The headline's own fields: executed-FLOP roofline of the step, the igemm aggregate, per-kernel rooflines, the other
    configs measured live, the full-frame protocol, the CPU baseline."""
    from understanding_flow_robustness_amd.patch_attack import PatchAttackStep
    net, args, step, batches = ctx["net"], ctx["args"], ctx["step"], ctx["batches"]
    band = getattr(step, "band", None) if step.cone is not None else None
    gflop = gflop_per_pair_step(step.win_hw if step.cone is not None else None, mc,
                                band.width if band is not None and band.width else None,
                                bool(band is not None and band.inc_layers))
    tf = gflop * B_PER_GPU / ms                        # per-GPU TFLOP/s (GFLOP/ms), executed work only
    line["config"].update({
                   "patch": "ONE 51x51 circular patch in patch coordinates, shown by every pair at its own placement",
                   "calls": f"attack() calls of max_count={mc} iterations, new frames + placement per call",
                   "prefix": (f"conv1-3 on a {step.win_hw[0]}x{step.win_hw[1]} window per pair"
                              if step.cone is not None else "full frame"),
                   "head_adjoint": (f"conv3_1/4/4_1/5 data gradients on a {band.width}-pixel column band"
                                    if band is not None else "full width"),
                   "loss": "cosine", "lr": 1000.0,
                   "parallelism": (f"dp{world}: pairs sharded, per-rank crop of the pre-clamp gradient to [3,51,51], "
                                   f"all-gather of {world} x 31 KB rows + fixed-order sum before the clamp")})
    line["roofline"] = {"bound": "mfma", "kernel": "attack step = 1 hipGraph launch", "achieved": round(tf, 2),
                     "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / PEAK_FP32_TFLOPS, 4),
                     "traffic": step_traffic(), "algorithmic_gflop_per_launch": round(gflop * B_PER_GPU, 1),
                     "full_frame_gflop_per_launch": round(GFLOP_PER_PAIR_STEP * B_PER_GPU, 1),
                     "peak_note": "the fp32 VECTOR / fp32-MFMA peak, kept as round 1's yardstick: the convolutions run as "
                                  "six bf16 MFMA products per float32 product (ceiling 416.7, see the top-level roofline), "
                                  "so this fraction may pass 1"}
    engine_on = bool(getattr(net, "__dict__", {}).get("_ufr_head_engines"))
    line["config"]["arithmetic"] = (
        "float32 end to end; the head's convolutions compute each float32 product on the bf16 matrix cores as three bf16 "
        "planes per operand and the six leading products, float32 accumulation (csrc/igemm.hip: error vs float64 at "
        "MIOpen's own fp32 level, tests/test_igemm_gpu.py) -- the head, conv1 / conv2 / conv3 of the full-frame prefix and of "
        "the window and their data gradients, the cost volume and both of its adjoints, predict_flow; no vendor "
        "convolution kernel runs in an iteration" if engine_on else
        "float32 end to end on MIOpen (UFR_ENGINE=0)")
    if world == 1:
        kernels, agg = kernel_rooflines(step, device, mc)
        step_line = line["roofline"]
        if agg is not None:
            # the dominant kernel: csrc/igemm.hip over the launches of one average iteration (live HIP-event timings)
            tf_i = agg["gflop"] / agg["ms"]
            line["roofline"] = {"bound": "mfma", "kernel": "igemm (csrc/igemm.hip), all launches of one iteration",
                                "achieved": round(tf_i, 1), "peak": round(PEAK_SPLIT6_TFLOPS, 1), "unit": "TFLOP/s",
                                "frac": round(tf_i / PEAK_SPLIT6_TFLOPS, 3),
                                "traffic": _pmc(IGEMM_TRAFFIC).get("per_iteration_bytes"),
                                "algorithmic_bytes_per_iteration": round(agg["bytes"]),
                                "ms_per_iteration": round(agg["ms"], 3), "algorithmic_gflop_per_iteration": round(agg["gflop"], 1),
                                "peak_note": "fp32-equivalent: 2.5 PFLOP/s dense bf16 MFMA / 6 products per float32 product",
                                "step": step_line}
        # the clock the igemm ran at, in this process: a slow box and a slow build can be told apart from this one line
        eng = next((e for e in getattr(net, "__dict__", {}).get("_ufr_head_engines", {}).values()
                    if e.B == B_PER_GPU and e.H == H and e.W == W), None)
        if eng is not None and agg is not None:
            mhz, tf_probe = measured_clock(eng, device)
            if mhz:
                line["roofline"]["clock_mhz"] = mhz
                line["roofline"]["nominal_clock_mhz"] = NOMINAL_MHZ
                line["roofline"]["frac_at_measured_clock"] = round(line["roofline"]["frac"] * NOMINAL_MHZ / mhz, 3)
                line["roofline"]["conv3_1_fwd_sustained_tflops"] = tf_probe
        line["roofline"]["kernels"] = kernels
        if not opt.no_other_configs:
            # the other BASELINE configs, one GPU's share each, measured live in this very process (steady-state inner
            # loop of their own step + the rooflines of their igemm launches): C4 PWC-Net, C3 RAFT with alt_cuda_corr,
            # C5 FlowNet2's universal-perturbation step (tools/bench_configs.py measures more: all-pairs RAFT, 8-pair RAFT)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_configs
            # (C3 at one pair -- the reference's batch size -- and at 8 pairs behind one patch: SURVEY.md 8d "C3 = 1...8")
            others = [(w, bench_configs.measure(w, 6)) for w in ("c4", "c3alt", "c3altb8", "c3", "c5")]
            for w, r in others:            # flat scalars first (the driver's parser keeps scalars in `config`, not nested values)
                line["config"][f"{w}_ms"] = r["ms_per_iteration"]
                if r.get("roofline"):
                    line["config"][f"{w}_igemm_frac"] = r["roofline"]["frac"]
            # RAFT's opt-in reduced precision as a SECOND C3 figure (flownets/raft.py: args.mixed_precision AND UFR_RAFT_PRECISION=bf16 --
            # one bf16 product per float32 product in the encoders and the update block, the reference's autocast; the C3 line above is float32)
            saved = os.environ.get("UFR_RAFT_PRECISION")
            os.environ["UFR_RAFT_PRECISION"] = "bf16"
            try:
                low = bench_configs.measure("c3alt", 6)
            finally:
                if saved is None:
                    os.environ.pop("UFR_RAFT_PRECISION", None)
                else:
                    os.environ["UFR_RAFT_PRECISION"] = saved
            line["config"]["c3alt_bf16_ms"] = low["ms_per_iteration"]
            line["config"]["other_configs"] = [r for _, r in others] + [low]
        if step.cone is not None and not opt.no_full_frame:
            # the same protocol with every frame-sized shortcut off (UFR_CONE=0): what the windowed prefix,
            # band and incremental forward are worth, measured in this very process
            ref_step = PatchAttackStep(net, args, B_PER_GPU, H, W, device=device, shared_patch=True,
                                       use_graph=not opt.no_graph, warmup=2, use_cone=False, patch_hw=(PATCH, PATCH))
            ref_load = lambda c: ref_step.load(*batches[c % 2]["args"], origins=batches[c % 2]["origins"])
            ref_load(0)
            ref_step.run(0)
            k = 6
            for c in range(2):
                ref_load(c); ref_step.enqueue(mc)
            torch.cuda.synchronize(device)
            t1 = time.perf_counter()
            done = 0
            while done < k:
                ref_load(done // mc); ref_step.enqueue(min(mc, k - done)); done += mc
            torch.cuda.synchronize(device)
            line["config"]["full_frame_attack_iters_per_s"] = round(B_PER_GPU * k / (time.perf_counter() - t1), 2)
        if not opt.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
            line["config"]["c1_cpu_forward_s"] = line["cpu_baseline"].get("c1_cpu_forward_s")


if __name__ == "__main__":
    main()

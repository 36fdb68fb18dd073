"""HBM-side traffic of one attack iteration from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame
    python tools/pmc_step_traffic.py /tmp/pf/*/*counter_collection.csv /tmp/pw/*/*counter_collection.csv 4

Sums the counter over the dispatches of the last `iters` iterations (delimited by gate_kernel, the last kernel
of every iteration; the per-call load() kernels fall inside and are averaged in), per iteration.
Units and the gfx950 correction follow MI355X_MICROARCH.md: both counters are in kilobytes, FETCH_SIZE is
doubled."""
import csv
import json
import sys
from collections import defaultdict


MARKER = ["gate_kernel"]


def per_iteration(path, counter, iters):
    rows = [r for r in csv.DictReader(open(path)) if r.get("Counter_Name", r.get("counter_name")) == counter]
    key = lambda r, *names: next(r[n] for n in names if n in r)
    rows.sort(key=lambda r: int(key(r, "Dispatch_Id", "dispatch_id")))
    names = [key(r, "Kernel_Name", "kernel_name") for r in rows]
    vals = [float(key(r, "Counter_Value", "counter_value")) for r in rows]
    marks = [i for i, n in enumerate(names) if MARKER[0] in n]
    if len(marks) < iters + 1:
        raise SystemExit(f"{path}: only {len(marks)} iterations recorded")
    lo, hi = marks[-iters - 1] + 1, marks[-1] + 1
    per_kernel = defaultdict(float)
    for n, v in zip(names[lo:hi], vals[lo:hi]):
        per_kernel[n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:60]] += v / iters
    return sum(vals[lo:hi]) / iters, per_kernel


# kernels launched once per iteration whose per-launch traffic bench.py quotes beside its live timing
SINGLE = {"corr_fwd_planes": "corr_planes_traffic.json", "corr_bwd_window_mfma_kernel": "corr_window_traffic.json",
          "altcorr_mfma_fwd": "altcorr_fwd_traffic.json", "altcorr_mfma_bwd2": "altcorr_bwd2_traffic.json"}


def main(fetch_csv, write_csv, iters, igemm_out=None, single_dir=None):
    f_kb, f_k = per_iteration(fetch_csv, "FETCH_SIZE", iters)
    w_kb, w_k = per_iteration(write_csv, "WRITE_SIZE", iters)
    if single_dir:
        for pat, name in SINGLE.items():
            fk = sum(v for k, v in f_k.items() if pat in k)
            wk = sum(v for k, v in w_k.items() if pat in k)
            if fk or wk:
                with open(f"{single_dir}/{name}", "w") as f:
                    json.dump({"_provenance": f"{pat}: the same two rocprofv3 --pmc passes as the step traffic file, one launch per "
                                              "iteration; KB -> bytes, FETCH_SIZE x2 per MI355X_MICROARCH.md",
                               "fetch_bytes": round(2.0 * fk * 1024), "write_bytes": round(wk * 1024),
                               "traffic_bytes": round(2.0 * fk * 1024 + wk * 1024)}, f, indent=1)
    if igemm_out:
        # the dominant kernel class alone (csrc/igemm.hip, both staging variants, + its split-K reduce)
        fi = sum(v for k, v in f_k.items() if "igemm" in k)
        wi = sum(v for k, v in w_k.items() if "igemm" in k)
        with open(igemm_out, "w") as f:
            json.dump({"_provenance": "igemm_* dispatches of the same two rocprofv3 --pmc passes as the step traffic file, per "
                                      "iteration; KB -> bytes, FETCH_SIZE x2 per MI355X_MICROARCH.md",
                       "fetch_bytes": round(2.0 * fi * 1024), "write_bytes": round(wi * 1024),
                       "per_iteration_bytes": round(2.0 * fi * 1024 + wi * 1024)}, f, indent=1)
    fetch_b, write_b = 2.0 * f_kb * 1024.0, w_kb * 1024.0              # FETCH_SIZE doubled on gfx950
    out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes with --kernel-trace only, over "
                          "bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-full-frame; mean over the last %d iterations "
                          "(load() of every second iteration included); KB -> bytes, FETCH_SIZE x2 per "
                          "MI355X_MICROARCH.md" % iters,
           "fetch_bytes_per_iteration": round(fetch_b), "write_bytes_per_iteration": round(write_b),
           "traffic_bytes_per_iteration": round(fetch_b + write_b),
           "top_fetch_kernels_MB": {k: round(2.0 * v * 1024 / 1e6, 1) for k, v in sorted(f_k.items(), key=lambda kv: -kv[1])[:8]},
           "top_write_kernels_MB": {k: round(v * 1024 / 1e6, 1) for k, v in sorted(w_k.items(), key=lambda kv: -kv[1])[:8]}}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if len(sys.argv) > 6:
        MARKER[0] = sys.argv[6]          # the last kernel of an iteration (universal_update_kernel for the perturbation step)
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 4, sys.argv[4] if len(sys.argv) > 4 else None,
         sys.argv[5] if len(sys.argv) > 5 else None)

"""Per-iteration kernel table from a rocprofv3 --kernel-trace CSV.

MIOpen's algorithm search and the eager warm-up dominate a whole-process --stats summary, so this
tool keeps only the steady state: the dispatches between the last `iters`+1 occurrences of the
marker kernel (default gate_kernel, the last kernel of every attack iteration).
    python tools/summarize_trace.py <kernel_trace.csv> <iters> [marker] > profiles/xxx.md
(trace bench.py with --no-cpu-baseline --no-full-frame so that the last iterations are the windowed step's)
"""
import csv
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"(?:_ZN2ck\w*?kernel_|ck::tensor_operation::device::kernel_)(\w+?)(?:INS_|<)", name)
    if m:
        return "ck::" + m.group(1)[:60]
    name = re.sub(r"\(.*$", "", name)
    return name[:100]


def main(path, iters, marker="gate_kernel"):
    rows = list(csv.DictReader(open(path)))
    ncol = next(c for c in rows[0] if c.lower() in ("kernel_name", "name"))
    scol = next(c for c in rows[0] if c.lower().startswith("start"))
    ecol = next(c for c in rows[0] if c.lower().startswith("end"))
    rows.sort(key=lambda r: int(r[scol]))
    marks = [i for i, r in enumerate(rows) if marker in r[ncol]]
    if len(marks) < iters + 1:
        raise SystemExit(f"only {len(marks)} '{marker}' dispatches in the trace")
    lo, hi = marks[-iters - 1] + 1, marks[-1] + 1
    sel = rows[lo:hi]
    wall = (int(sel[-1][ecol]) - int(sel[0][scol])) / 1e6
    agg = OrderedDict()
    for r in sel:
        k = short(r[ncol])
        a = agg.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += int(r[ecol]) - int(r[scol])
    busy = sum(v[1] for v in agg.values()) / 1e6
    print(f"steady state: last {iters} iterations, {len(sel)} dispatches, wall {wall / iters:.3f} ms/iteration, "
          f"sum of kernel durations {busy / iters:.3f} ms/iteration\n")
    print("| kernel | calls/iter | avg us | ms/iter | % of kernel time |")
    print("|---|---|---|---|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {c / iters:.1f} | {t / c / 1e3:.1f} | {t / 1e6 / iters:.3f} | {100 * t / 1e6 / busy:.1f} |")
    # the last iteration, dispatch by dispatch (which launch of a kernel is the slow one)
    last = rows[marks[-2] + 1:marks[-1] + 1]
    print(f"\nlast iteration, in order ({len(last)} dispatches): kernel, us")
    for r in last:
        print(f"{short(r[ncol])[:60]:60s} {(int(r[ecol]) - int(r[scol])) / 1e3:8.1f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), sys.argv[3] if len(sys.argv) > 3 else "gate_kernel")

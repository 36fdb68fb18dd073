"""Condense a rocprofv3 *_kernel_stats.csv into a short per-step table (profiles/*.md)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    m = re.match(r"ck::tensor_operation::device::kernel_(\w+)<", name)
    if m:
        return "ck::" + m.group(1) + "<...>"
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


def main(path, iters, top=40):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print(f"# {path}: {len(rows)} kernels, total GPU time {tot / 1e6:.1f} ms over {iters} iterations "
          f"= {tot / 1e6 / iters:.3f} ms/iteration\n")
    print("| kernel | calls | calls/iter | avg us | ms/iter | % |")
    print("|---|---|---|---|---|---|")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| `{k}` | {c} | {c / iters:.1f} | {t / c / 1e3:.1f} | {t / 1e6 / iters:.3f} | {100 * t / tot:.1f} |")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 40)

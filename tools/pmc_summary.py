"""Average rocprofv3 --pmc counters per kernel (counter_collection.csv)."""
import csv, sys, re
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "corr_"
kcol = next(c for c in rows[0] if c.lower() == "kernel_name")
ccol = next(c for c in rows[0] if c.lower() == "counter_name")
vcol = next(c for c in rows[0] if c.lower() == "counter_value")
agg = defaultdict(lambda: defaultdict(list))
for r in rows:
    if pat in r[kcol]:
        name = re.sub(r"\(.*$", "", r[kcol].replace("void (anonymous namespace)::", ""))
        agg[name][r[ccol]].append(float(r[vcol]))
for k, cs in agg.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}   (n={len(v)})")

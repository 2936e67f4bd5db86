"""Print per-kernel averages of every counter in a rocprofv3 --pmc csv directory:  python tools/pmc_sum.py DIR [name-filter]"""
import csv, glob, os, sys
from collections import defaultdict
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        k = r["Kernel_Name"].split("(")[0]
        c = agg[k][r["Counter_Name"]]
        c[0] += 1; c[1] += float(r["Counter_Value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k, cs in agg.items():
    if flt in k:
        print(k[:80], "  ".join(f"{n}={v[1] / v[0]:.4g}" for n, v in sorted(cs.items())))

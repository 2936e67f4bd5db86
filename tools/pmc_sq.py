"""Per-kernel sums of every counter in a rocprofv3 --pmc output directory: python tools/pmc_sq.py <dir> [name filter]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f, newline="")):
        name = row.get("Kernel_Name", "?").split("(")[0]
        if flt in name:
            acc[name][row["Counter_Name"]] += float(row["Counter_Value"]); n[name].add(row.get("Dispatch_Id"))
for name, c in acc.items():
    k = len(n[name])
    print(name[:70], "launches", k)
    for cn, v in sorted(c.items()):
        print(f"   {cn:32s} {v / k:16.0f} per launch")

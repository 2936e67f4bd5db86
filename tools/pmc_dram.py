"""Where do the fabric reads of a kernel end up — in DRAM or in the Infinity Cache?  

    python tools/pmc_dram.py <read_pass_dir> [<write_pass_dir>] [out.json] [per_gpu_batch]

read pass : rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace -- python3 bench.py ...
write pass: rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-trace -- python3 bench.py ...

FETCH_SIZE / WRITE_SIZE (tools/pmc_traffic.py) are derived from the L2's memory-side request counters TCC_EA0_RDREQ / WRREQ and
cannot tell an Infinity-Cache hit from a DRAM access (MI355X_MICROARCH.md §HBM).  TCC_EA0_RDREQ_DRAM / WRREQ_DRAM count the
requests "destined for DRAM (MC)"; their share of all requests of a kernel, applied to that kernel's (corrected) FETCH /
WRITE bytes, splits the fabric traffic into a DRAM part and the rest.  Request sizes are not assumed: only ratios are used.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_counters(dirname, names):
    per_kernel = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(int)
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {dirname}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                c = row.get("Counter_Name")
                if c not in names:
                    continue
                k = row.get("Kernel_Name", "?").split("(")[0]
                per_kernel[k][c] += float(row["Counter_Value"])
                if c == names[0]:
                    launches[k] += 1
    return per_kernel, launches


def main():
    args = [a for a in sys.argv[1:]]
    rd_dir = args[0]
    wr_dir = args[1] if len(args) > 1 and os.path.isdir(args[1]) else None
    rest = args[2 if wr_dir else 1:]
    out_path = rest[0] if rest else None
    batch = int(rest[1]) if len(rest) > 1 else None
    rd, nrd = read_counters(rd_dir, ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_RDREQ_32B_sum"])
    wr, nwr = (read_counters(wr_dir, ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_WRREQ_64B_sum"]) if wr_dir else ({}, {}))
    out = {}
    for k in sorted(set(rd) | set(wr)):
        r, w = rd.get(k, {}), wr.get(k, {})
        d = {"launches": nrd.get(k, 0) or nwr.get(k, 0)}
        if r:
            tot = r.get("TCC_EA0_RDREQ_sum", 0.0)
            d.update(rdreq_per_launch=tot / max(nrd.get(k, 1), 1), rdreq_dram_frac=(r.get("TCC_EA0_RDREQ_DRAM_sum", 0.0) / tot) if tot else None,
                     rdreq_32b_frac=(r.get("TCC_EA0_RDREQ_32B_sum", 0.0) / tot) if tot else None)
        if w:
            tot = w.get("TCC_EA0_WRREQ_sum", 0.0)
            d.update(wrreq_per_launch=tot / max(nwr.get(k, 1), 1), wrreq_dram_frac=(w.get("TCC_EA0_WRREQ_DRAM_sum", 0.0) / tot) if tot else None,
                     wrreq_64b_frac=(w.get("TCC_EA0_WRREQ_64B_sum", 0.0) / tot) if tot else None)
        out[k] = d
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from clibd_amd.build import csrc_hash

    text = json.dumps({"unit": "requests per launch; *_dram_frac = share of the L2's memory-side requests destined for DRAM (the rest: Infinity Cache / other)",
                       "csrc_sha16": csrc_hash(), "per_gpu_batch": batch, "kernels": out}, indent=1)
    if out_path:
        with open(out_path, "w") as fh:
            fh.write(text + "\n")
    rows = sorted(out.items(), key=lambda kv: -(kv[1].get("rdreq_per_launch", 0.0) * max(kv[1]["launches"], 1)))
    for k, d in rows[:30]:
        f = lambda v: "   -  " if v is None else f"{v:6.3f}"
        print(f"{k[:72]:72s} n={d['launches']:5d} rd/launch={d.get('rdreq_per_launch', 0.0) / 1e6:9.2f} M dram={f(d.get('rdreq_dram_frac'))} "
              f"wr/launch={d.get('wrreq_per_launch', 0.0) / 1e6:9.2f} M dram={f(d.get('wrreq_dram_frac'))}")


if __name__ == "__main__":
    main()

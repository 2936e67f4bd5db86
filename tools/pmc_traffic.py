"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, MI355X_MICROARCH.md §HBM) into per-kernel HBM
bytes per launch.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> [out.json] [per_gpu_batch]

The output records the hash of clibd_amd/csrc (clibd_amd.build.csrc_hash) and the per-GPU batch of the profiled run, so that
bench.py only quotes it for the code and workload it was measured on.

Corrections applied (the guide's gfx950 notes): FETCH_SIZE is reported in KB and tallies 128-B requests at 64 B for wide
(16 B/lane) streaming reads -> bytes = FETCH_SIZE * 1024 * 2; WRITE_SIZE is reported in KB and is exact for 16 B/lane
stores -> bytes = WRITE_SIZE * 1024.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def read_counter(dirname, counter):
    per_kernel = defaultdict(lambda: [0, 0.0])
    files = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {dirname}")
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "?")
                name = name.split("(")[0]
                per_kernel[name][0] += 1
                per_kernel[name][1] += float(row["Counter_Value"])
    return per_kernel


def main():
    fetch_dir, write_dir = sys.argv[1], sys.argv[2]
    fetch = read_counter(fetch_dir, "FETCH_SIZE")
    write = read_counter(write_dir, "WRITE_SIZE")
    out = {}
    for name in sorted(set(fetch) | set(write)):
        nf, vf = fetch.get(name, [0, 0.0])
        nw, vw = write.get(name, [0, 0.0])
        rd = vf * 1024.0 * 2.0 / max(nf, 1)
        wr = vw * 1024.0 / max(nw, 1)
        out[name] = {"launches_fetch_pass": nf, "launches_write_pass": nw, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr,
                     "hbm_bytes_per_launch": rd + wr}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from clibd_amd.build import csrc_hash

    text = json.dumps({"unit": "bytes per launch (FETCH_SIZE KB x 1024 x 2 [gfx950 half-count correction] + WRITE_SIZE KB x 1024)",
                       "csrc_sha16": csrc_hash(), "per_gpu_batch": int(sys.argv[4]) if len(sys.argv) > 4 else None,
                       "kernels": out}, indent=1)
    if len(sys.argv) > 3:
        with open(sys.argv[3], "w") as fh:
            fh.write(text + "\n")
    rows = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * max(kv[1]["launches_fetch_pass"], 1))
    for name, d in rows[:25]:
        print(f"{name[:70]:70s} n={d['launches_fetch_pass']:5d} rd={d['read_bytes_per_launch'] / 1e6:9.2f} MB wr={d['write_bytes_per_launch'] / 1e6:9.2f} MB")


if __name__ == "__main__":
    main()

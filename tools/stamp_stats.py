"""Copy a rocprofv3 --stats kernel summary out of its output directory with the hash of the kernel sources it measured.

    python tools/stamp_stats.py <rocprofv3 -d directory> <destination.csv> "<command that was profiled>"

The destination is the *_kernel_stats.csv as rocprofv3 wrote it plus ONE trailing row whose Name is
`[meta] csrc_hash=<clibd_amd.build.csrc_hash()> command=<...>` and whose numeric columns are 0 (the kernel-stats
files must say which tree they are of, as the PMC JSONs do; `bench.py` quotes PMC traffic only on a hash match)."""
import csv
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clibd_amd.build import csrc_hash  # noqa: E402


def main():
    src_dir, dst, cmd = sys.argv[1], sys.argv[2], (sys.argv[3] if len(sys.argv) > 3 else "")
    cands = sorted(glob.glob(os.path.join(src_dir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getsize, reverse=True)
    if not cands:
        print(f"stamp_stats: no *kernel_stats.csv under {src_dir}", file=sys.stderr)
        return 1
    rows = list(csv.reader(open(cands[0])))
    width = len(rows[0])
    rows.append([f"[meta] csrc_hash={csrc_hash()} command={cmd}"] + ["0"] * (width - 1))
    with open(dst, "w", newline="") as f:
        csv.writer(f, quoting=csv.QUOTE_ALL).writerows(rows)
    print(f"stamp_stats: {cands[0]} -> {dst} (csrc {csrc_hash()})")
    return 0


if __name__ == "__main__":
    sys.exit(main())

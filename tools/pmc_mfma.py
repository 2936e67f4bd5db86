"""Summarise a rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE pass into per-kernel
MFMA utilisation (north_star: "rocprof ... MFMA utilisation against chip peak").

    python tools/pmc_mfma.py <pmc_dir> [out.json]

Per MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32), summed over
the chip's SIMDs as rocprofv3 reports it; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel's active cycles on one XCD
= GRBM_GUI_ACTIVE / 8 and   mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                name = row.get("Kernel_Name", "?").split("(")[0]
                acc[name][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[name].add(row.get("Dispatch_Id"))
    out = {}
    for name, c in acc.items():
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        out[name] = {"launches": len(launches[name]), "SQ_VALU_MFMA_BUSY_CYCLES": mfma, "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES", 0.0),
                     "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES", 0.0), "GRBM_GUI_ACTIVE": gui,
                     "mfma_util": (mfma / (gui / 8.0 * 1024.0)) if gui > 0 else None}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from clibd_amd.build import csrc_hash

    text = json.dumps({"formula": "mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)", "csrc_sha16": csrc_hash(),
                       "kernels": out}, indent=1)
    if len(sys.argv) > 2:
        with open(sys.argv[2], "w") as fh:
            fh.write(text + "\n")
    rows = sorted(out.items(), key=lambda kv: -kv[1]["GRBM_GUI_ACTIVE"])
    for name, v in rows[:25]:
        u = v["mfma_util"]
        print(f"{name[:80]:80s} n={v['launches']:4d} gui={v['GRBM_GUI_ACTIVE']:.3e} mfma_busy={v['SQ_VALU_MFMA_BUSY_CYCLES']:.3e} util={'-' if u is None else f'{u:.3f}'}")


if __name__ == "__main__":
    main()

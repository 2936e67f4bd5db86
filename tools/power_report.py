"""Joins tools/power_sampler.py traces with the windows printed by tools/bench_gemm_shapes.py and the bench's step run.

    python tools/power_report.py <dir written by tools/gpu_power_probe.sh>
"""
import json, re, sys
import numpy as np


def load(path):
    P = np.loadtxt(path, comments="#")
    ncard = (P.shape[1] - 1) // 2
    k = int(np.argmax([np.nanmean(P[:, 1 + 2 * i]) for i in range(ncard)]))
    return P[:, 0], P[:, 1 + 2 * k], P[:, 2 + 2 * k], k, ncard


def main():
    d = sys.argv[1].rstrip("/") + "/"
    t, pw, clk, k, n = load(d + "power_gemm.txt")
    print(f"card column {k} of {n} (the one under load)")
    for line in open(d + "shapes.log"):
        m = re.search(r"^(\S+).*sustained\s+([\d.]+) us\s+(\d+) TF window ([\d.]+) ([\d.]+)", line)
        if not m:
            continue
        t0, t1 = float(m.group(4)), float(m.group(5))
        s = (t > t0 + 1.0) & (t < t1 - 0.2)
        f = np.nanmean(clk[s])
        tf = float(m.group(3))
        print(f"{m.group(1):14s} {tf:5.0f} TF  power {pw[s].mean():6.0f} W ({pw[s].min():.0f}..{pw[s].max():.0f})  sclk {f:5.0f} MHz ({np.nanmin(clk[s]):.0f}..{np.nanmax(clk[s]):.0f})"
              f"  MFMA peak at that clock {2.5 * f / 2400:5.2f} PF -> {tf / (2500 * f / 2400):.2f} of it")
    t, pw, clk, k, n = load(d + "power_step.txt")
    t0 = float(open(d + "step_t0.txt").read()); t1 = float(open(d + "step_t1.txt").read())
    b = json.loads(open(d + "bench.json").read().strip().splitlines()[-1])
    busy = (t > t0) & (t < t1) & (pw > 600)
    print(f"training step: {b['ms_per_step']:.1f} ms, {b['value']:.0f} pairs/s, GEMM {b['roofline']['achieved']:.0f} TF; while busy (n={busy.sum()}): "
          f"power {pw[busy].mean():.0f} W (p10 {np.percentile(pw[busy], 10):.0f}, p90 {np.percentile(pw[busy], 90):.0f}), "
          f"sclk {np.nanmean(clk[busy]):.0f} MHz (p10 {np.nanpercentile(clk[busy], 10):.0f}, p90 {np.nanpercentile(clk[busy], 90):.0f})")


if __name__ == "__main__":
    main()

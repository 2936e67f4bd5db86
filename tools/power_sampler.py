"""Board power / shader clock sampler (sysfs hwmon of every card, read-only) — one line per sample: t then "power_W sclk_MHz" per card.
The card under load is the one whose power moves; tools that read the file pick the column pair with the largest mean power.

    python tools/power_sampler.py <out.txt> <seconds> [period_s]
"""
import glob, sys, time


def cards():
    out = []
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for pn in ("power1_input", "power1_average"):
            try:
                float(open(f"{d}/{pn}").read())
                out.append((d, pn))
                break
            except Exception:
                continue
    return out


def rd(path):
    try:
        return float(open(path).read()) / 1e6
    except Exception:
        return float("nan")


def main():
    out, secs = sys.argv[1], float(sys.argv[2])
    period = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    cs = cards()
    t_end = time.time() + secs
    with open(out, "w") as f:
        f.write("# cards: " + " ".join(d for d, _ in cs) + "\n")
        while time.time() < t_end:
            t = time.time()
            vals = []
            for d, pn in cs:
                vals.append(f"{rd(f'{d}/{pn}'):.1f} {rd(f'{d}/freq1_input'):.0f}")
            f.write(f"{t:.3f} " + " ".join(vals) + "\n"); f.flush()
            time.sleep(period)


if __name__ == "__main__":
    main()

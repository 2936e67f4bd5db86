"""kernel-trace CSV -> how much of each kernel class's time overlapped with launches of another class."""
import csv, sys, glob
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f))]
cls = lambda n: "gemm" if "gemm256" in n else ("ln" if "layernorm" in n else None)
ev = [(cls(n), s, e) for n, s, e in rows if cls(n)]
g = sorted((s, e) for c, s, e in ev if c == "gemm"); l = sorted((s, e) for c, s, e in ev if c == "ln")
def overlap(a, b):
    tot = 0; j = 0
    for s, e in a:
        for s2, e2 in b:
            if e2 <= s: continue
            if s2 >= e: break
            tot += min(e, e2) - max(s, s2)
    return tot
print("gemm total", sum(e - s for s, e in g) / 1e6, "ms; ln total", sum(e - s for s, e in l) / 1e6, "ms; overlapped", overlap(g, l) / 1e6, "ms")
print("gemm mean", sum(e - s for s, e in g) / len(g) / 1e3, "us; ln mean", sum(e - s for s, e in l) / len(l) / 1e3, "us")

"""Busy / idle analysis of ONE training step in a rocprofv3 --kernel-trace csv (step = between two adamw_kernel launches).
usage: python tools/trace_gaps.py <dir with *_kernel_trace.csv> [step index from the end, default 2]"""
import csv, glob, os, sys
rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f, newline="")):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:60], r.get("Stream_Id", "?")))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
i0, i1 = ad[-k - 1], ad[-k]
step = rows[i0 + 1 : i1 + 1]
span = step[-1][1] - rows[i0][1]
busy = 0; cur_s, cur_e = step[0][0], step[0][1]
gaps = [(step[0][0] - rows[i0][1], step[0][2])]
for s, e, n, q in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(e - s for s, e, _, _ in step)
streams = {}
for s, e, n, q in step:
    streams[q] = streams.get(q, 0) + (e - s)
print(f"step window {span / 1e6:.2f} ms, {len(step)} kernels, union-busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), sum of durations {tot / 1e6:.2f} ms")
print("per stream busy ms:", {q: round(v / 1e6, 2) for q, v in streams.items()})
gaps.sort(reverse=True)
print("largest idle gaps (us, next kernel):", [(round(g / 1e3, 1), n[:28]) for g, n in gaps[:10]])
print("idle total %.2f ms in %d gaps" % (sum(g for g, _ in gaps) / 1e6, len(gaps)))

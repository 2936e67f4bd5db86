#!/bin/bash
# Kernel traces of the step with and without its collectives over a one-rank RCCL group; idle-gap analysis.  usage: bash tools/gpu_rccl_trace.sh <tag>
set -u
OUT=gpurun_out/${1:-rct}
mkdir -p "$OUT"
export TMPDIR=/tmp
for mode in 0 1; do
  export CLIBD_FORCE_COLLECTIVES=$mode
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace$mode" -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d > "$OUT/trace$mode.log" 2>&1
  echo "mode $mode exit $?"
  python tools/trace_gaps.py "$OUT/trace$mode" 2 > "$OUT/gaps$mode.txt" 2>&1; cat "$OUT/gaps$mode.txt"
  python - "$OUT/trace$mode" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("  kernel time total %.1f ms over the run; nccl/rccl kernels:" % (sum(float(r["TotalDurationNs"]) for r in rows) / 1e6))
for r in rows:
    if "ccl" in r["Name"].lower():
        print("   ", r["Name"][:90], r["Calls"], "calls", round(float(r["AverageNs"]) / 1e3, 1), "us avg")
PY
  find "$OUT/trace$mode" -name "*kernel_trace.csv" -size +20M -delete; find "$OUT/trace$mode" -name "*.db" -delete
done

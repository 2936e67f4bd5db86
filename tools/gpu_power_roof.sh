#!/bin/bash
# The board's delivered rates at its power cap: bare MFMA, MFMA + LDS, copy, read, write, MFMA + copy.  usage: bash tools/gpu_power_roof.sh <tag>
set -u
TAG=${1:-pr}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python tools/power_sampler.py "$OUT/power_gemm.txt" 60 0.05 &
S1=$!
timeout -k 10 120 tools/micro/power_roof 4 > "$OUT/shapes.log" 2>&1
kill $S1 2>/dev/null; wait $S1 2>/dev/null
cat "$OUT/shapes.log"

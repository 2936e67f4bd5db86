#!/bin/bash
# Same-box A/B of library variants built by tools/build_variant.sh: GEMM shapes (sustained rates) with board power sampled beside them.
# usage: bash tools/gpu_gemm_ab.sh <tag> "<variant names>" [seconds per shape]
set -u
TAG=${1:-ab}
VARS=${2:-"base"}
SECS=${3:-1.5}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
python tools/power_sampler.py "$OUT/power.txt" 600 0.05 &
SMP=$!
for v in $VARS; do
  CLIBD_HIP_LIB=$PWD/build_ab/lib_$v.so timeout 300 python tools/bench_gemm_shapes.py 403456 6 $SECS > "$OUT/shapes_$v.log" 2>&1
  echo "== $v"; grep -E "sum|TF" "$OUT/shapes_$v.log" | sed -e 's/window.*//' | cut -c1-110
done
kill $SMP 2>/dev/null; wait $SMP 2>/dev/null

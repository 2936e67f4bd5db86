#!/bin/bash
# round 5: the fp8-forward lines beside the bf16 one on ONE box, after the parity tests of the touched kernels
# usage (GPU box, repo root): bash tools/gpu_r5_fp8.sh [tag] ["modes", default "bf16 pooled all"]
set -u
OUT=gpurun_out/${1:-r5f}
MODES=${2:-"bf16 pooled all"}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_fp8_gpu.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest_fp8.log" 2>&1
echo "pytest exit $?"; tail -3 "$OUT/pytest_fp8.log"
for tw in $MODES; do
  if [ "$tw" = "bf16" ]; then FLAG=""; else FLAG="--fp8-forward $tw"; fi
  timeout -k 10 300 python bench.py $FLAG --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics --gemm-breakdown > "$OUT/bench_${tw}_b2048.json" 2> "$OUT/bench_${tw}_b2048.err" && echo "$tw ok" && head -c 300 "$OUT/bench_${tw}_b2048.json" && echo
  timeout -k 10 300 python bench.py $FLAG --per-gpu-batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics > "$OUT/bench_${tw}_b1024.json" 2> "$OUT/bench_${tw}_b1024.err" && echo "$tw b1024 ok" && head -c 300 "$OUT/bench_${tw}_b1024.json" && echo
done

#!/bin/bash
# A/B of two library builds on one box: build_ab/libclibd_old.so (CLIBD_HIP_LIB) against the in-tree build.
# usage: bash tools/gpu_gemm_ab.sh <tag>
set -u
TAG=${1:-ab}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
timeout 600 python -m pytest tests/test_ops_gpu.py -q -x -k "gemm" -p no:cacheprovider > "$OUT/pytest_gemm.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gemm.log"; tail -3 "$OUT/pytest_gemm.log"
for rep in 1 2; do
  CLIBD_HIP_LIB=$PWD/build_ab/libclibd_old.so timeout 300 python tools/bench_gemm_shapes.py > "$OUT/shapes_old_$rep.log" 2>&1
  timeout 300 python tools/bench_gemm_shapes.py > "$OUT/shapes_new_$rep.log" 2>&1
done
paste "$OUT/shapes_old_1.log" "$OUT/shapes_new_1.log" | cut -c1-140
paste "$OUT/shapes_old_2.log" "$OUT/shapes_new_2.log" | cut -c1-140

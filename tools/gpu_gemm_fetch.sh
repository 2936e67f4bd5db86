#!/bin/bash
# Fabric read traffic (FETCH_SIZE) of the GEMM shapes under library variants (tools/build_variant.sh).  usage: bash tools/gpu_gemm_fetch.sh <tag> "<variants>"
set -u
TAG=${1:-gf}
VARS=${2:-"base"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
for v in $VARS; do
  export CLIBD_HIP_LIB=$PWD/build_ab/lib_$v.so
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$v" -- python3 tools/bench_gemm_shapes.py 403456 2 > "$OUT/fetch_$v.log" 2>&1
  echo "$v exit $?"
  find "$OUT/fetch_$v" -name "*.db" -delete
done

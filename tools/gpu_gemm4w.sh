#!/bin/bash
# the four-wave x 512-register 256x256 GEMM against the product's eight-wave kernel, one process, interleaved
# usage (GPU box, repo root): bash tools/gpu_gemm4w.sh [tag] [seconds per arm]
set -u
OUT=gpurun_out/${1:-r4g}
mkdir -p "$OUT"
cd tools/micro
[ -x ./gemm4w ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip -ldl
timeout -k 10 500 ./gemm4w ../../clibd_amd/libclibd_hip.so ${2:-2} > "../../$OUT/gemm4w.log" 2>&1
echo "gemm4w exit $?"; cat "../../$OUT/gemm4w.log"

#!/bin/bash
# VERDICT r4 item 1 (step A / B): the split-role 256x128 GEMM workgroup (4 compute + 4 memory waves) against the product's eight-wave
# 256x256 kernel, one process, interleaved.   usage (GPU box, repo root): bash tools/gpu_gemmsr.sh [tag] [seconds per arm]
set -u
OUT=gpurun_out/${1:-r5g}
mkdir -p "$OUT"
cd tools/micro
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o gemmsr gemmsr.hip -ldl || exit 1
timeout -k 10 500 ./gemmsr ../../clibd_amd/libclibd_hip.so ${2:-2} > "../../$OUT/gemmsr.log" 2>&1
echo "gemmsr exit $?"; cat "../../$OUT/gemmsr.log"

#!/bin/bash
# Where a per-GPU-batch-256 step (the 8-GPU configuration's share) spends its time: kernel stats (towers serialized and overlapped), idle gaps.
set -u
OUT=gpurun_out/${1:-b256p}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python bench.py --per-gpu-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --gemm-breakdown > "$OUT/bench.json" 2> "$OUT/bench.err"
grep "^\[gemm\]" "$OUT/bench.err" | head -20
export CLIBD_TOWER_STREAMS=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/serial" -- python3 bench.py --per-gpu-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d > "$OUT/serial.log" 2>&1
unset CLIBD_TOWER_STREAMS
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/streams" -- python3 bench.py --per-gpu-batch 256 --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d > "$OUT/streams.log" 2>&1
for m in serial streams; do echo "== $m"; python tools/trace_gaps.py "$OUT/$m" 2; done
find "$OUT" -name "*.db" -delete

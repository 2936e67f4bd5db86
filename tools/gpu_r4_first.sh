#!/bin/bash
# round 4, first box: the GPU suite with the new bench-path tests, the bench line (host_enqueue_ms), the per-rank shape of configs[2], host profile
set -u
OUT=gpurun_out/r4a
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --gemm-breakdown > "$OUT/bench_b2048.json" 2> "$OUT/bench_b2048.err" && echo "bench ok" && tail -c 1200 "$OUT/bench_b2048.json"
timeout -k 10 200 python bench.py --per-gpu-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > "$OUT/bench_b256.json" 2> "$OUT/bench_b256.err" && echo "b256 ok" && tail -c 900 "$OUT/bench_b256.json"
timeout -k 10 200 python tools/host_profile.py 256 > "$OUT/host_profile.txt" 2>&1; head -3 "$OUT/host_profile.txt"

#!/bin/bash
# quick loop for the single-pass attention backward: op-level parity, then kernel timings
set -u
OUT=gpurun_out/${1:-attnq}
mkdir -p "$OUT"
timeout -k 10 300 python -m pytest tests/test_ops_gpu.py -q -x -k "single_pass" -p no:cacheprovider > "$OUT/pytest_sp.log" 2>&1
rc=$?; echo "sp op tests exit $rc"; tail -3 "$OUT/pytest_sp.log"
[ $rc -ne 0 ] && exit 1
timeout -k 10 300 python tools/bench_attn.py 256 2048 > "$OUT/bench_attn.log" 2>&1; grep "^B=" "$OUT/bench_attn.log"

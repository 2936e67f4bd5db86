#!/bin/bash
# round 5: the adapters' backward with a partials workspace (deterministic, no contended float atomics) — parity tests, op timing, step A/B is in the bench lines
set -u
OUT=gpurun_out/${1:-r5w}
mkdir -p "$OUT"
timeout -k 10 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q -x -k "lora or tower_parity or full_step or trajectory or ranks" -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -2 "$OUT/pytest.log"
timeout -k 10 300 python tools/bench_lora.py 2>&1 | grep "^M=.*lora_backward" | tee "$OUT/bench_lora.log"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics > "$OUT/bench_b2048.json" 2> "$OUT/bench_b2048.err" && head -c 200 "$OUT/bench_b2048.json" && echo
timeout -k 10 200 python bench.py --per-gpu-batch 256 --steps 30 --warmup 5 --no-cpu-baseline --no-h2d --no-ref-numerics > "$OUT/bench_b256.json" 2> "$OUT/bench_b256.err" && head -c 200 "$OUT/bench_b256.json" && echo

#!/bin/bash
# What a 1-GPU box can say about the 8-GPU configuration (per-GPU batch 256): host enqueue headroom, the step with its collectives over a
# one-rank RCCL group, and the full fine-tune step with its bucketed all-reduce issued from the backward.  usage: bash tools/gpu_w8_rehearsal.sh <tag>
set -u
OUT=gpurun_out/${1:-w8}
mkdir -p "$OUT"
OMP_NUM_THREADS=1 timeout 300 python tools/host_enqueue_time.py 256 2>&1 | grep "b=" | tee "$OUT/host_enqueue.txt"
for mode in 0 1; do
  env CLIBD_FORCE_COLLECTIVES=$mode timeout 600 python bench.py --per-gpu-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-gemm-timing > "$OUT/b256_force$mode.json" 2> "$OUT/b256_force$mode.err"
  env CLIBD_FORCE_COLLECTIVES=$mode timeout 600 python bench.py --full-finetune --per-gpu-batch 256 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-gemm-timing > "$OUT/fullft_force$mode.json" 2> "$OUT/fullft_force$mode.err"
  for n in b256 fullft; do python - "$OUT/${n}_force$mode.json" "$n force=$mode" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:18s} {d['ms_per_step']:.2f} ms/step {d['value']:.0f} pairs/s loss {d['loss']:.5f}")
PY
  done
done

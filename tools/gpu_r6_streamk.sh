#!/bin/bash
# Round 6: the stream-K tail of the 256x256 GEMM (CLIBD_GEMM_STREAMK=1 default, =0 off), in-step, interleaved on one box: per-GPU batch 256 (the 8-GPU per-rank shape),
# 512, 1024 and the metric's 2048.
set -u
OUT=gpurun_out/${1:-r6sk}; mkdir -p "$OUT"
for B in ${SK_BATCHES:-256 2048 512 1024}; do
  echo "== per-GPU batch $B ==" >> "$OUT/streamk_ab.log"
  for v in 1 0 1 0 1 0; do
    ST=10; [ $B -le 512 ] && ST=20
    CLIBD_GEMM_STREAMK=$v timeout -k 10 300 python bench.py --per-gpu-batch $B --steps $ST --warmup 4 --no-cpu-baseline --no-h2d --no-ref-numerics --no-configs4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('streamk=$v b=$B', round(d['ms_per_step'],3), 'ms  ', round(d['value'],1), 'pairs/s   GEMM frac', round(r['frac'],4), ' gemm ms', round(r['gemm_ms_per_step'],2))" >> "$OUT/streamk_ab.log"
  done
done
cat "$OUT/streamk_ab.log"

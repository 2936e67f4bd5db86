#!/bin/bash
# In-step A/B of the 256x256 GEMM's tile order: m-bands of 4 (rounds 1-4) against W-stationary n-groups of 6 (variant libraries built by tools/build_variant.sh with
# -DCLIBD_WS_MIN_TILES_N=<n>: the W-stationary order for launches with at least n column tiles; 100000 = never = "band", 1 = always = "wsall").
# Round 3 measured the orders on isolated GEMM launches (same time, 15-25 % fewer fabric reads); the two-stream step shares the fabric between the towers.
# usage: bash tools/gpu_ws_step_ab.sh <tag> "<variants: product | band | ws9 | wsall ...>" ["bench flags"]
set -u
OUT=gpurun_out/${1:-wsab}; mkdir -p "$OUT"
FLAGS=${3:-""}
for v in ${2:-"product band wsall product band wsall"}; do
  if [ "$v" != "product" ]; then export CLIBD_HIP_LIB=build_ab/lib_$v.so; else unset CLIBD_HIP_LIB; fi
  timeout -k 10 300 python bench.py $FLAGS --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics --no-configs4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', '$FLAGS', round(d['ms_per_step'],2), 'ms   GEMM frac', round(r['frac'],4), ' board W', round(r.get('board',{}).get('board_power_w') or 0), 'sclk', round(r.get('board',{}).get('sclk_mhz') or 0))"
done

#!/bin/bash
# Round-6 experiment session 2: (a) bound of an 8-bit QKV dgrad, (b) balanced persistent grids (-DCLIBD_BALANCED_GRID) in-step at b = 256 and b = 2048,
# (c) DRAM share of the LayerNorm kernels' memory-side requests at b = 256 and b = 2048, (d) full fine-tune: the bench line + its configs4 record (bf16 vs the 8-bit pooled dgrad).
set -u
OUT=gpurun_out/${1:-r6exp2}; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 300 python tools/bench_qkv_dgrad8_bound.py 2048 1024 > "$OUT/qkv_dgrad8_bound.log" 2>&1; cat "$OUT/qkv_dgrad8_bound.log"
echo "== balanced grids, in-step, b = 256 (20 steps) ==" > "$OUT/balgrid_ab.log"
for v in product balgrid product balgrid product balgrid; do
  if [ "$v" != "product" ]; then export CLIBD_HIP_LIB=build_ab/lib_$v.so; else unset CLIBD_HIP_LIB; fi
  timeout -k 10 300 python bench.py --per-gpu-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-h2d --no-ref-numerics --no-configs4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v b256', round(d['ms_per_step'],3), 'ms   GEMM frac', round(r['frac'],4))" >> "$OUT/balgrid_ab.log"
done
echo "== balanced grids, in-step, b = 2048 ==" >> "$OUT/balgrid_ab.log"
for v in product balgrid product balgrid; do
  if [ "$v" != "product" ]; then export CLIBD_HIP_LIB=build_ab/lib_$v.so; else unset CLIBD_HIP_LIB; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics --no-configs4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v b2048', round(d['ms_per_step'],2), 'ms   GEMM frac', round(r['frac'],4))" >> "$OUT/balgrid_ab.log"
done
unset CLIBD_HIP_LIB
cat "$OUT/balgrid_ab.log"
for B in 256 2048; do
  PMC_BATCH=$B bash tools/gpu_round.sh "${1:-r6exp2}/dram_b$B" "dram" > "$OUT/dram_b$B.out" 2>&1
  grep -i "layernorm" "$OUT/dram_b$B/pmc_dram.txt" | head -12
done
timeout 600 python bench.py --full-finetune --steps 5 --warmup 2 --no-cpu-baseline --no-h2d --gemm-breakdown > "$OUT/bench_fullft_b2048.json" 2> "$OUT/bench_fullft_b2048.err"
echo "fullft exit $?"; python -c "
import json; d=json.load(open('$OUT/bench_fullft_b2048.json')); print('fullft', d['ms_per_step'], d['value'], d['roofline']['step_frac']); print(json.dumps(d.get('configs4'))[:1500])"
du -sh "$OUT"

#!/bin/bash
# round 4, third box: the GPU suite (fp8 tower selection, full fine-tune on the bf16 residual-gradient stream), then the secondary lines
set -u
OUT=gpurun_out/r4c
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider -s > "$OUT/pytest_gpu.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log"; grep "^\[fp8\|^\[BERT" "$OUT/pytest_gpu.log"
for rg in bf16 fp32; do
  CLIBD_RESIDUAL_GRAD=$rg timeout -k 10 200 python bench.py --full-finetune --steps 6 --warmup 2 --no-cpu-baseline --no-h2d > "$OUT/bench_fullft_b2048_$rg.json" 2> "$OUT/bench_fullft_b2048_$rg.err" && echo "fullft $rg ok" && head -c 330 "$OUT/bench_fullft_b2048_$rg.json" && echo
done
for tw in pooled all; do
  timeout -k 10 200 python bench.py --fp8-forward $tw --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench_fp8_${tw}_b2048.json" 2> "$OUT/bench_fp8_${tw}_b2048.err" && echo "fp8 $tw ok" && head -c 330 "$OUT/bench_fp8_${tw}_b2048.json" && echo
  timeout -k 10 200 python bench.py --fp8-forward $tw --per-gpu-batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench_fp8_${tw}_b1024.json" 2> "$OUT/bench_fp8_${tw}_b1024.err" && echo "fp8 $tw b1024 ok" && head -c 330 "$OUT/bench_fp8_${tw}_b1024.json" && echo
done
timeout -k 10 200 python bench.py --per-gpu-batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench_bf16_b1024.json" 2> "$OUT/bench_bf16_b1024.err" && echo "bf16 b1024 ok" && head -c 330 "$OUT/bench_bf16_b1024.json" && echo
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench_b2048.json" 2> "$OUT/bench_b2048.err" && echo "bench ok" && head -c 330 "$OUT/bench_b2048.json" && echo
du -sh "$OUT"

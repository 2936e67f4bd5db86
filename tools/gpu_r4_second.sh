#!/bin/bash
# round 4, second box: three-wave attention kernels (tests + A/B), full fine-tune re-profile
set -u
OUT=gpurun_out/r4b
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "attention" -p no:cacheprovider > "$OUT/pytest_attn.log" 2>&1
echo "pytest exit $?" >> "$OUT/pytest_attn.log"; tail -4 "$OUT/pytest_attn.log"
for w in 4 3; do
  CLIBD_ATTN_BWD_WAVES=$w CLIBD_ATTN_FWD_WAVES=$w timeout -k 10 200 python tools/bench_attn.py 256 2048 > "$OUT/bench_attn_w$w.log" 2>&1
  echo "== waves $w"; grep "S=133" "$OUT/bench_attn_w$w.log"
done
timeout -k 10 300 python bench.py --full-finetune --steps 6 --warmup 2 --no-cpu-baseline --no-h2d --gemm-breakdown > "$OUT/bench_fullft_b2048.json" 2> "$OUT/bench_fullft_b2048.err" && echo "fullft ok" && tail -c 700 "$OUT/bench_fullft_b2048.json"
export CLIBD_TOWER_STREAMS=0
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_fullft" -- python3 bench.py --full-finetune --steps 3 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d > "$OUT/prof_fullft.log" 2>&1
echo "prof exit $?"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"

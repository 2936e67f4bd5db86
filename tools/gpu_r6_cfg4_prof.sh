#!/bin/bash
# Round 6: rocprofv3 kernel stats (towers on one stream) of BASELINE configs[4]'s per-GPU batch 1024 in bf16 and in the three fp8 modes of bench.py's
# `configs4` record.  usage (repo root, GPU box): bash tools/gpu_r6_cfg4_prof.sh <tag>
set -u
OUT=gpurun_out/${1:-cfg4prof}; mkdir -p "$OUT"
export TMPDIR=/tmp CLIBD_TOWER_STREAMS=0
COMMON="--per-gpu-batch 1024 --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4"
run() {   # name, extra flags
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_$1" -- python3 bench.py $COMMON $2 > "$OUT/prof_$1.log" 2>&1
  echo "prof $1 exit $?"
  python tools/stamp_stats.py "$OUT/prof_$1" "$OUT/kernel_stats_b1024_$1.csv" "CLIBD_TOWER_STREAMS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py $COMMON $2"
}
run bf16 ""
run ffn_dgrad_pooled "--fp8-forward pooled_ffn --dgrad fp8-pooled"
run dgrad_all "--dgrad fp8"
run ffn_dgrad_all "--fp8-forward pooled_ffn --dgrad fp8"
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"

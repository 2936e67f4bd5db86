#!/bin/bash
# round 5: the 8-bit dgrad (numerics dgrad = fp8) beside the bf16 dgrad, with and without the fp8 forward selections, on ONE box
# usage (GPU box, repo root): bash tools/gpu_r5_dgrad.sh [tag] ["configs", default below]   config = <forward>:<dgrad>, forward in bf16|pooled|pooled_mlp|all
set -u
OUT=gpurun_out/${1:-r5dg}
CFGS=${2:-"bf16:bf16 bf16:fp8 pooled:bf16 pooled:fp8 bf16:bf16"}
B=${3:-2048}
mkdir -p "$OUT"
export TMPDIR=/tmp
i=0
for cfg in $CFGS; do
  fw=${cfg%%:*}; dg=${cfg##*:}; i=$((i+1))
  if [ "$fw" = "bf16" ]; then FLAG=""; else FLAG="--fp8-forward $fw"; fi
  name="${i}_${fw}_dgrad_${dg}_b${B}"
  timeout -k 10 300 python bench.py $FLAG --dgrad $dg --per-gpu-batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics --gemm-breakdown > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.err" \
    && python - "$OUT/bench_$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:36s} {d['ms_per_step']:8.2f} ms  {d['value']:8.1f} pairs/s  loss {d.get('loss')}  frac {d['roofline']['frac']:.3f}  sclk {d['roofline'].get('board', {}).get('sclk_mhz')}")
PY
done
if [ "${4:-}" = "prof" ]; then   # kernel statistics of the pooled forward + 8-bit dgrad step, serial towers
  export CLIBD_TOWER_STREAMS=0
  CMD="python3 bench.py --fp8-forward pooled --dgrad fp8 --per-gpu-batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics"
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_dgrad8" -- $CMD > "$OUT/prof_dgrad8.log" 2>&1
  echo "prof exit $?"
  unset CLIBD_TOWER_STREAMS
  python tools/stamp_stats.py "$OUT/prof_dgrad8" "$OUT/kernel_stats_pooled_dgrad8_serial.csv" "CLIBD_TOWER_STREAMS=0 rocprofv3 --kernel-trace --stats -- $CMD"
  find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
  find "$OUT" -name "*.db" -delete
fi

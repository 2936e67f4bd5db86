#!/bin/bash
# The step with its collectives over a ONE-rank RCCL group against the local step, alternating on one box.  usage: bash tools/gpu_rccl_w1.sh <tag>
set -u
OUT=gpurun_out/${1:-rccl}
mkdir -p "$OUT"
for mode in 0 1 0 1; do
  env CLIBD_FORCE_COLLECTIVES=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-gemm-timing > "$OUT/bench_force$mode.json" 2> "$OUT/bench_force$mode.err"
  echo "force=$mode exit $?"; python - "$OUT/bench_force$mode.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(f"  {d['ms_per_step']:.2f} ms/step {d['value']:.0f} pairs/s loss {d['loss']:.5f} {d.get('collectives', '')[:60]}")
PY
done

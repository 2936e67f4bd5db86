#!/bin/bash
# A/B of a two-valued environment knob on one box: gradient budget against the oracle, the GPU suite (default value), step time alternating.
# usage: bash tools/gpu_knob_ab.sh <tag> <KNOB> <reference value> <new value> [notest]
set -u
OUT=gpurun_out/${1:-knob}; KNOB=$2; A=$3; B=$4; NOTEST=${5:-}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python tools/residual_grad_budget.py $KNOB $A $B > "$OUT/grad_budget.log" 2>&1; echo "budget exit $?"; cat "$OUT/grad_budget.log"
if [ -z "$NOTEST" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?"; tail -15 "$OUT/pytest_gpu.log"
fi
for mode in $A $B $A $B; do
  env $KNOB=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --gemm-breakdown > "$OUT/bench_$mode.json" 2> "$OUT/bench_$mode.err"
  echo "bench $mode exit $?"; python - "$OUT/bench_$mode.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]; b = r.get("board") or {}
print(f"  {d['ms_per_step']:.1f} ms/step {d['value']:.0f} pairs/s loss {d['loss']:.4f} gemm {r['gemm_ms_per_step']:.1f} ms {r['achieved']:.0f} TF  board {b.get('board_power_w', 0):.0f} W {b.get('sclk_mhz', 0):.0f} MHz")
PY
done

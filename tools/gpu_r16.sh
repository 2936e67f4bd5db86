#!/bin/bash
# A/B of the bf16 residual-gradient stream on one box: gradient budget, parity tests, step time in both modes.
set -u
OUT=gpurun_out/${1:-r16}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 600 python tools/residual_grad_budget.py > "$OUT/grad_budget.log" 2>&1; echo "budget exit $?"; cat "$OUT/grad_budget.log"
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?"; tail -15 "$OUT/pytest_gpu.log"
for mode in fp32 bf16 fp32 bf16; do
  CLIBD_RESIDUAL_GRAD=$mode timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --gemm-breakdown > "$OUT/bench_$mode.json" 2> "$OUT/bench_$mode.err"
  echo "bench $mode exit $?"; python - "$OUT/bench_$mode.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"  {d['ms_per_step']:.1f} ms/step {d['value']:.0f} pairs/s loss {d['loss']:.4f} gemm {r['gemm_ms_per_step']:.1f} ms {r['achieved']:.0f} TF")
PY
done

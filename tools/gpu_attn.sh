#!/bin/bash
# Single-pass attention backward: op-level parity first, then the model suites, then the step time with both backward kernels.
set -u
OUT=gpurun_out/${1:-attn}
mkdir -p "$OUT"
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -q -x -k "attention" -p no:cacheprovider > "$OUT/pytest_attn.log" 2>&1
rc=$?; echo "attention op tests exit $rc"; tail -25 "$OUT/pytest_attn.log"
[ $rc -ne 0 ] && exit 1
timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=20 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1; echo "pytest exit $?"; tail -12 "$OUT/pytest_gpu.log"
for mode in 2phase sp 2phase sp; do
  CLIBD_ATTN_BWD=$mode timeout -k 10 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench_$mode.json" 2> "$OUT/bench_$mode.err"
  echo "bench $mode exit $?"; python - "$OUT/bench_$mode.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(f"  {d['ms_per_step']:.1f} ms/step {d['value']:.0f} pairs/s loss {d['loss']:.4f} gemm {r['gemm_ms_per_step']:.1f} ms {r['achieved']:.0f} TF")
PY
done
python tools/bench_attn.py > "$OUT/bench_attn.log" 2>&1; tail -20 "$OUT/bench_attn.log"

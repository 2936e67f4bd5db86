#!/bin/bash
# round 5: the step with numerics ln_fold off / on (CLIBD_LN_FOLD), alternating on one box, at the metric's batch and at per-GPU batch 256
set -u
OUT=gpurun_out/${1:-r5v}
mkdir -p "$OUT"
for rep in 1 2; do
  for mode in off on; do
    CLIBD_LN_FOLD=$mode timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics > "$OUT/b2048_$mode$rep.json" 2> "$OUT/b2048_$mode$rep.err" && python - "$OUT/b2048_$mode$rep.json" "b2048 ln_fold=$mode rep $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'pairs/s loss', round(d['loss'],5), 'gemm', round(r['gemm_ms_per_step'],2), 'ms frac', round(r['frac'],4), d['config']['numerics']['image_encoder']['ln_fold'])
PY
    CLIBD_LN_FOLD=$mode timeout -k 10 200 python bench.py --per-gpu-batch 256 --steps 30 --warmup 5 --no-cpu-baseline --no-h2d --no-ref-numerics > "$OUT/b256_$mode$rep.json" 2> "$OUT/b256_$mode$rep.err" && python - "$OUT/b256_$mode$rep.json" "b256 ln_fold=$mode rep $rep" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[2], round(d['ms_per_step'],2), 'ms', round(d['value'],1), 'pairs/s loss', round(d['loss'],5))
PY
  done
done

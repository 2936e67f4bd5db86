#!/bin/bash
# round 5: bench.py's contract test with the new legs, and the N-rank code path of those legs (two ranks sharing the one GPU over gloo: a code-path check)
set -u
OUT=gpurun_out/${1:-r5c}
mkdir -p "$OUT"
timeout -k 10 600 python -m pytest tests/test_bench_contract_gpu.py -m gpu -q -x -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest exit $?"; tail -2 "$OUT/pytest.log"
CLIBD_BENCH_SHARED_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --global-batch 512 --steps 3 --warmup 1 --no-cpu-baseline > "$OUT/bench_w2_shared.json" 2> "$OUT/bench_w2_shared.err"
echo "w2 exit $?"; python - "$OUT/bench_w2_shared.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','n_gpus','ms_per_step','invalid','loss')}); print('ref', d['reference_numerics']['value'], 'h2d', d['h2d_inclusive']['value'], d['h2d_inclusive']['uint8_images']['value'])
PY

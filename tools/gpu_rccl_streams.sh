#!/bin/bash
# Why the towers stop overlapping once a process group exists: step time with the collectives forced, under candidate remedies.
set -u
OUT=gpurun_out/${1:-rcs}
mkdir -p "$OUT"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 600 python bench.py ${CLIBD_BENCH_EXTRA:-} --steps 8 --warmup 3 --no-cpu-baseline --no-h2d --no-gemm-timing > "$OUT/$name.json" 2> "$OUT/$name.err"
  python - "$OUT/$name.json" "$name" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"{sys.argv[2]:28s} {d['ms_per_step']:.2f} ms/step {d['value']:.0f} pairs/s loss {d['loss']:.5f}")
PY
}
run local CLIBD_FORCE_COLLECTIVES=0
run forced CLIBD_FORCE_COLLECTIVES=1
run forced_prio_high CLIBD_FORCE_COLLECTIVES=1 CLIBD_TOWER_STREAM_PRIORITY=-1
run local_prio_high CLIBD_FORCE_COLLECTIVES=0 CLIBD_TOWER_STREAM_PRIORITY=-1
run forced_hwq8 CLIBD_FORCE_COLLECTIVES=1 GPU_MAX_HW_QUEUES=8
CLIBD_BENCH_EXTRA="--per-gpu-batch 256" run b256_local CLIBD_FORCE_COLLECTIVES=0
CLIBD_BENCH_EXTRA="--per-gpu-batch 256" run b256_forced CLIBD_FORCE_COLLECTIVES=1
CLIBD_BENCH_EXTRA="--per-gpu-batch 256" run b256_forced_prio CLIBD_FORCE_COLLECTIVES=1 CLIBD_TOWER_STREAM_PRIORITY=-1
CLIBD_BENCH_EXTRA="--per-gpu-batch 256" run b256_forced_hwq8 CLIBD_FORCE_COLLECTIVES=1 GPU_MAX_HW_QUEUES=8

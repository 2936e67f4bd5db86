#!/bin/bash
# Board power and shader clock while (1) each GEMM shape runs for ~4 s, (2) the training step runs.  usage: bash tools/gpu_power_probe.sh <tag>
set -u
TAG=${1:-pw}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
rocm-smi --showmaxpower --showpower --showclocks > "$OUT/smi_idle.txt" 2>&1
ls /sys/class/drm/card*/device/hwmon/hwmon*/ > "$OUT/hwmon_ls.txt" 2>&1
python tools/power_sampler.py "$OUT/power_gemm.txt" 70 0.05 &
S1=$!
timeout 300 python tools/bench_gemm_shapes.py 403456 8 4 > "$OUT/shapes.log" 2>&1
kill $S1 2>/dev/null; wait $S1 2>/dev/null
cat "$OUT/shapes.log"
python tools/power_sampler.py "$OUT/power_step.txt" 100 0.05 &
S2=$!
date +%s.%N > "$OUT/step_t0.txt"
timeout 600 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench.json" 2> "$OUT/bench.err"
date +%s.%N > "$OUT/step_t1.txt"
kill $S2 2>/dev/null; wait $S2 2>/dev/null
tail -c 400 "$OUT/bench.json"
wc -l "$OUT/power_gemm.txt" "$OUT/power_step.txt"

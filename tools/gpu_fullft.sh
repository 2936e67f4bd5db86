#!/bin/bash
# full fine-tune (model_config.disable_lora) profile at a given per-GPU batch
set -u
B=${1:-256}; OUT=gpurun_out/fullft_b$B; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python bench.py --full-finetune --per-gpu-batch $B --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --gemm-breakdown > $OUT/bench.json 2> $OUT/bench.err
tail -c 900 $OUT/bench.json; echo
export CLIBD_TOWER_STREAMS=0
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 bench.py --full-finetune --per-gpu-batch $B --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d > $OUT/prof.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +20M -delete; find $OUT -name "*.db" -delete
python - <<PY
import csv,glob
f=glob.glob('$OUT/prof_serial/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f))); steps=7
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("serial kernel ms/step", tot/steps/1e6)
for r in rows[:28]:
    t=float(r['TotalDurationNs']); n=r['Name'].replace('void clibd::','').replace('clibd::','').split('(')[0][:72]
    print(f"{n:74s} n={int(r['Calls']):5d} {t/steps/1e6:8.2f} ms/step avg {float(r['AverageNs'])/1e3:9.1f} us {100*t/tot:5.1f}%")
PY

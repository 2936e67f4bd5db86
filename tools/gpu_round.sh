#!/bin/bash
# One GPU-box session: parity tests, the bench line on the metric's configuration, secondary configs, rocprofv3 summaries.
# usage (from the repo root on the GPU box): bash tools/gpu_round.sh <tag> [steps to run, default "test bench b256 tri prof"; also: fp8 pmc dram eval]
set -u
TAG=${1:-run}
WHAT=${2:-"test bench b256 tri prof"}
OUT=gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
has() { [[ " $WHAT " == *" $1 "* ]]; }
if has test; then
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 -p no:cacheprovider > "$OUT/pytest_gpu.log" 2>&1
  echo "pytest exit $?" >> "$OUT/pytest_gpu.log"; tail -5 "$OUT/pytest_gpu.log"
fi
if has bench; then
  timeout 900 python bench.py --steps 10 --warmup 3 --gemm-breakdown > "$OUT/bench_b2048.json" 2> "$OUT/bench_b2048.err"
  echo "bench exit $?"; tail -c 1500 "$OUT/bench_b2048.json"
fi
if has b256; then
  timeout 600 python bench.py --per-gpu-batch 256 --steps 20 --warmup 5 --no-cpu-baseline --no-configs4 --gemm-breakdown > "$OUT/bench_b256.json" 2> "$OUT/bench_b256.err"
  echo "b256 exit $?"; tail -c 600 "$OUT/bench_b256.json"
fi
if has tri; then
  timeout 600 python bench.py --tri-modal --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_trimodal_b2048.json" 2> "$OUT/bench_trimodal.err"
  echo "tri exit $?"; tail -c 600 "$OUT/bench_trimodal_b2048.json"
fi
if has fp8; then   # BASELINE configs[4]: opt-in fp8-forward mode — the training-grade selection (mean-pooled towers) and the embedding-grade one (all)
  for tw in pooled all; do
    timeout 600 python bench.py --fp8-forward $tw --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --gemm-breakdown > "$OUT/bench_fp8_${tw}_b2048.json" 2> "$OUT/bench_fp8_${tw}_b2048.err"
    echo "fp8 $tw exit $?"; tail -c 400 "$OUT/bench_fp8_${tw}_b2048.json"
    timeout 600 python bench.py --fp8-forward $tw --per-gpu-batch 1024 --steps 10 --warmup 3 --no-cpu-baseline --no-h2d > "$OUT/bench_fp8_${tw}_b1024.json" 2> "$OUT/bench_fp8_${tw}_b1024.err"
    echo "fp8 $tw b1024 exit $?"; tail -c 400 "$OUT/bench_fp8_${tw}_b1024.json"
  done
fi
if has prof; then
  export CLIBD_TOWER_STREAMS=0
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_serial" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/prof_serial.log" 2>&1
  echo "prof serial exit $?"
  unset CLIBD_TOWER_STREAMS
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_streams" -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/prof_streams.log" 2>&1
  echo "prof streams exit $?"
  python tools/stamp_stats.py "$OUT/prof_serial" "$OUT/kernel_stats_serial.csv" "CLIBD_TOWER_STREAMS=0 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4"
  python tools/stamp_stats.py "$OUT/prof_streams" "$OUT/kernel_stats_two_streams.csv" "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4"
  find "$OUT" -name "*kernel_trace.csv" -size +20M -delete   # keep the stats, drop oversized raw traces
  find "$OUT" -name "*.db" -delete
fi
if has pmc; then
  B=${PMC_BATCH:-2048}
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch" -- python3 bench.py --per-gpu-batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/pmc_fetch.log" 2>&1
  timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_write" -- python3 bench.py --per-gpu-batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/pmc_write.log" 2>&1
  python tools/pmc_traffic.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic_b$B.json" $B > "$OUT/pmc_traffic.txt" 2>&1
  timeout 900 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$OUT/pmc_mfma" -- python3 bench.py --per-gpu-batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/pmc_mfma.log" 2>&1
  python tools/pmc_mfma.py "$OUT/pmc_mfma" "$OUT/pmc_mfma_b$B.json" > "$OUT/pmc_mfma.txt" 2>&1
  find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
  find "$OUT" -name "*.db" -delete
  tail -30 "$OUT/pmc_traffic.txt"; tail -30 "$OUT/pmc_mfma.txt"
fi
if has dram; then   # DRAM share of the L2's memory-side requests per kernel (tools/pmc_dram.py)
  B=${PMC_BATCH:-2048}
  timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d "$OUT/pmc_dram_rd" -- python3 bench.py --per-gpu-batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/pmc_dram_rd.log" 2>&1
  echo "dram rd exit $?"
  timeout 900 rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d "$OUT/pmc_dram_wr" -- python3 bench.py --per-gpu-batch $B --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/pmc_dram_wr.log" 2>&1
  echo "dram wr exit $?"
  python tools/pmc_dram.py "$OUT/pmc_dram_rd" "$OUT/pmc_dram_wr" "$OUT/pmc_dram_b$B.json" $B > "$OUT/pmc_dram.txt" 2>&1
  find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
  find "$OUT" -name "*.db" -delete
  tail -32 "$OUT/pmc_dram.txt"
fi
if has eval; then
  timeout 900 python bench.py --eval --steps 5 --warmup 2 > "$OUT/bench_eval.json" 2> "$OUT/bench_eval.err"
  echo "eval exit $?"; tail -c 1800 "$OUT/bench_eval.json"
fi
du -sh "$OUT"

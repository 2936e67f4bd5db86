#!/bin/bash
# Round-6 experiment session 1 (variant libraries built on the CPU box by tools/build_variant.sh, selected with CLIBD_HIP_LIB):
#   ntaux : -DCLIBD_NT_AUX_LOADS          the MUL_AUX / ADD_AUX epilogues' aux stream loaded with the non-temporal hint (gemm256)
#   att3w : -DCLIBD_ATT_BWD_LONG_WAVES=3  the long-sequence attention backward capped at 168 registers = three waves per SIMD (attention)
# (a) attention kernels alone, (b) in-step A/B at b = 2048, interleaved, (c) FETCH_SIZE of the product and of the ntaux variant.
set -u
OUT=gpurun_out/${1:-r6exp1}; mkdir -p "$OUT"
export TMPDIR=/tmp
echo "== attention kernels (tools/bench_attn.py 2048) ==" > "$OUT/attn_ab.log"
for v in product att3w product att3w; do
  if [ "$v" != "product" ]; then export CLIBD_HIP_LIB=build_ab/lib_$v.so; else unset CLIBD_HIP_LIB; fi
  echo "-- $v" >> "$OUT/attn_ab.log"
  timeout -k 10 300 python tools/bench_attn.py 2048 2>/dev/null | grep "^B=" >> "$OUT/attn_ab.log"
done
unset CLIBD_HIP_LIB
cat "$OUT/attn_ab.log"
echo "== in-step A/B, b = 2048 ==" > "$OUT/step_ab.log"
bash tools/gpu_ws_step_ab.sh "${1:-r6exp1}/ab" "product ntaux att3w product ntaux att3w product ntaux" >> "$OUT/step_ab.log" 2>&1
cat "$OUT/step_ab.log"
# FETCH_SIZE per kernel: product, then the ntaux variant (one step each)
for v in product ntaux; do
  if [ "$v" != "product" ]; then export CLIBD_HIP_LIB=build_ab/lib_$v.so; else unset CLIBD_HIP_LIB; fi
  timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_fetch_$v" -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timing --no-h2d --no-ref-numerics --no-configs4 > "$OUT/pmc_fetch_$v.log" 2>&1
  echo "pmc fetch $v exit $?"
done
unset CLIBD_HIP_LIB
python - "$OUT" <<'EOF'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
for v in ("product", "ntaux"):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(out, f"pmc_fetch_{v}", "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == "FETCH_SIZE":
                    n = row["Kernel_Name"].split("(")[0]
                    per[n][0] += 1; per[n][1] += float(row["Counter_Value"])
    print(f"== FETCH_SIZE x 1024 x 2 per launch, {v} (gemm256 kernels)")
    for n, (c, s) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        if "gemm256" in n:
            print(f"  {n[:110]:110s} launches {c:4d}  {s * 2048 / c / 1e9:7.3f} GB/launch  {s * 2048 / 1e9:8.1f} GB total")
    print(f"  all kernels: {sum(s for _, s in per.values()) * 2048 / 1e9:.1f} GB")
EOF
find "$OUT" -name "*kernel_trace.csv" -size +20M -delete
find "$OUT" -name "*.db" -delete
du -sh "$OUT"

#!/bin/bash
# what the DNA attention backward's SECOND fetch of q, k, v, dO costs — the product library against a timing-only
# variant that takes everything from its LDS images (build_ab/lib_norefetch.so: tools/build_variant.sh norefetch "-DCLIBD_ATT_NO_REFETCH" attention),
# alternating in one session; then the fp8 gradient test with its cosines printed (pooled / pooled_mlp / all on MI355X).
set -u
OUT=gpurun_out/${1:-r5t}
mkdir -p "$OUT"
for rep in 1 2; do
  for lib in product norefetch; do
    if [ $lib = product ]; then unset CLIBD_HIP_LIB; else export CLIBD_HIP_LIB=$PWD/build_ab/lib_norefetch.so; fi
    timeout -k 10 200 python tools/bench_attn.py 2048 2>&1 | grep "^B=" | sed "s/^/[$lib $rep] /" | tee -a "$OUT/attn_refetch.log"
  done
done
unset CLIBD_HIP_LIB
timeout -k 10 600 python -m pytest tests/test_fp8_gpu.py -m gpu -q -x -s -k "spread or matches_the_fp8_oracle" -p no:cacheprovider 2>&1 | grep "^\[fp8\|passed\|failed" | tee "$OUT/fp8_cosines.log"

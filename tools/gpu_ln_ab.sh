#!/bin/bash
# LayerNorm kernel variants (tools/build_variant.sh <name> "<flags>" layernorm) on one box, alternating.  usage: bash tools/gpu_ln_ab.sh <tag> "<variants>" <bench script>
set -u
OUT=gpurun_out/${1:-lnab}; VARS=${2:-"lnb1 lnb2"}; SCRIPT=${3:-tools/bench_ln_bwd.py}
mkdir -p "$OUT"
for v in $VARS $VARS; do
  echo "== $v"; CLIBD_HIP_LIB=$PWD/build_ab/lib_$v.so timeout 300 python $SCRIPT 2>&1 | grep "M="
done

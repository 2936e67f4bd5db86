#!/bin/bash
# Experimental library variant for same-box A/B runs: build_ab/lib_<name>.so = the in-tree objects with ONE unit recompiled under extra -D flags.
# usage: bash tools/build_variant.sh <name> "<-D flags>" [unit, default gemm256]      (run python -m clibd_amd.build first)
set -e
NAME=$1; FLAGS=${2:-}; UNIT=${3:-gemm256}
mkdir -p build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -Wno-inline-asm $FLAGS -c clibd_amd/csrc/$UNIT.hip -o build_ab/${UNIT}_$NAME.o 2>/dev/null
OBJS=""
for n in capi gemm gemm256 gemm256_tn layernorm attention lora elementwise loss topk paramgrad; do
  if [ "$n" != "$UNIT" ]; then OBJS="$OBJS clibd_amd/csrc/build/$n.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_ab/lib_$NAME.so $OBJS build_ab/${UNIT}_$NAME.o 2>/dev/null
echo built build_ab/lib_$NAME.so

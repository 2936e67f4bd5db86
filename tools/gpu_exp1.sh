#!/bin/bash
set -u
OUT=gpurun_out/exp1; mkdir -p $OUT
timeout 600 python tools/diag_batch_split.py > $OUT/diag_batch_split.log 2>&1; tail -30 $OUT/diag_batch_split.log
python -m clibd_amd.build --diag > $OUT/build_diag.log 2>&1; tail -2 $OUT/build_diag.log
timeout 900 python tools/exp_gemm_knobs.py 403456 > $OUT/knobs_b2048.log 2>&1; cat $OUT/knobs_b2048.log
timeout 600 python tools/exp_gemm_knobs.py 50432 > $OUT/knobs_b256.log 2>&1; cat $OUT/knobs_b256.log

#!/bin/bash
# ring GEMM: requests of phase 3 moved to the front of the next phase 1 (variant library -DVAENPVC_NR_SCHED=1): parity, A/B
set -u
OUT=gpurun_out/r6c11; mkdir -p $OUT
bash scripts/ab_call.sh r6c11 -k "ring_gemm" -t enc4_fwd,heads_fwd,enc4_dgrad,enc3_fwd,merge_dgrad -r 3 -s 2 default lib:nrsched1

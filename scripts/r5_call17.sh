#!/bin/bash
set -u
OUT=gpurun_out/r5c17; mkdir -p $OUT
scripts/ab_env.sh 3 "VAENPVC_SIDE_STREAM=1" "-" 2>&1 | tee $OUT/ab_side.txt
for e in "VAENPVC_SIDE_STREAM=1" "VAENPVC_SIDE_STREAM=0"; do
  env $e python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e bf16', round(d['ms_per_step'],4))"
done 2>&1 | tee -a $OUT/ab_side.txt

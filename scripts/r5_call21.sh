#!/bin/bash
set -u
OUT=gpurun_out/r5c21; mkdir -p $OUT
for i in 1 2; do
  for e in "VAENPVC_FCR_SITES=0x08a" "VAENPVC_FCR_SITES=0x28a"; do
    env $e python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e bf16', round(d['ms_per_step'],4))"
  done
done 2>&1 | tee $OUT/ab_bf16.txt
VAENPVC_FCR_SITES=0x28a timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "bf16_mode" 2>&1 | tail -2

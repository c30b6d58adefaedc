#!/bin/bash
set -u
OUT=gpurun_out/r5c22; mkdir -p $OUT
python scripts/site_times.py --precision bf16 > $OUT/sites_bf16.txt 2>&1; tail -1 $OUT/sites_bf16.txt
for i in 1 2; do
  for e in "VAENPVC_X=0" "VAENPVC_CV_SITES=0xe28a" "VAENPVC_FC_SITES=0xfb1" "VAENPVC_TN_W4_TILES=99"; do
    env $e python bench.py --precision bf16 --steps 40 --warmup 10 --no-cpu-baseline --no-literal --no-modes --no-convert 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e bf16', round(d['ms_per_step'],4))"
  done
done 2>&1 | tee $OUT/ab_bf16.txt

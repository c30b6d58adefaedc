#!/bin/bash
set -u
OUT=gpurun_out/r5c23; mkdir -p $OUT
for i in 1 2 3 4 5; do
  rm -f gpurun_out/parity_report.txt
  timeout 600 python -m pytest tests/test_gpu_frame.py -x -q -m gpu -k "twenty" > $OUT/pytest_$i.log 2>&1
  tail -1 $OUT/pytest_$i.log
  grep -E 'layered-bf16x2 (per-step worst|  largest|sum-type)' gpurun_out/parity_report.txt | cut -c1-200
done

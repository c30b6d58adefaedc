#!/bin/bash
# GPU gate as the driver runs it (-x -q -m gpu), N times, + the small-batch noise-floor soak.  usage: scripts/gpu_gate.sh <tag> [N]
set -u
TAG=${1:-gate}; N=${2:-1}
OUT=gpurun_out/$TAG
mkdir -p $OUT
for i in $(seq 1 $N); do
  rm -f gpurun_out/parity_report.txt
  timeout 900 python -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest_$i.log 2>&1
  echo "pytest rc=$?" >> $OUT/pytest_$i.log
  cp gpurun_out/parity_report.txt $OUT/parity_report_$i.txt 2>/dev/null
  tail -2 $OUT/pytest_$i.log
done

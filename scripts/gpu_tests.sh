#!/bin/bash
# GPU test-suite only (quick gpurun call).  usage: scripts/gpu_tests.sh <tag> [pytest args]
set -u
TAG=${1:-t}; shift || true
OUT=gpurun_out/$TAG
mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 1500 python -m pytest tests -m gpu -q --timeout 600 "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
grep -E "passed|failed|FAILED" $OUT/pytest.log | tail -30

#!/bin/bash
set -u
OUT=gpurun_out/r5c20; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "fused_thin or ragged or benchmarked or all_tuned or decoder_tail" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log

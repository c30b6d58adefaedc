#!/bin/bash
set -u
OUT=gpurun_out/r6c12; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "layernorm_on_load or fixture or conversion" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
bash scripts/ab_call.sh r6c12 -t dec3_fwd,dec2_fwd,loss -r 3 -s 3 lib:lna3 default
bash scripts/ab_call.sh r6c12b -t "" -s 2 -m "--precision bf16" lib:lna3 default

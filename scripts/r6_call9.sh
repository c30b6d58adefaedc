#!/bin/bash
# encoder layer 3 forward on the ring: three frames per wave at a time in the frame pass, constants fetched before the K loop
set -u
OUT=gpurun_out/r6c9; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "ring_gemm or view_conv" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
bash scripts/ab_call.sh r6c9 -t enc3_fwd,enc4_fwd -r 2 -s 2 env:VAENPVC_CG_SF_RING=0 default

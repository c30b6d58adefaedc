#!/bin/bash
# LNA with the conversion in the MFMA shadow + lean OST + non-temporal plane stores + row-aligned merge epilogue: parity, then same-box A/Bs
set -u
OUT=gpurun_out/r6c3; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "layernorm_on_load or three_planes or fixture or plane_gemm or ragged_large or a_resident" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
bash scripts/ab_call.sh r6c3 -t dec2_fwd,dec2_stats_planes,dec3_fwd,loss,dec3_wgrad,merge_fwd -r 2 -s 2 env:VAENPVC_D2_LNA=0 default
bash scripts/ab_call.sh r6c3b -t "" -s 1 -m "--precision bf16" env:VAENPVC_D2_LNA=0 default
bash scripts/ab_call.sh r6c3c -t "" -s 2 -m "--precision bf16x3" default env:VAENPVC_FB_LAYERS=0 env:VAENPVC_FB_LAYERS=0,VAENPVC_SIDE_STREAM=0 env:VAENPVC_SIDE_STREAM=0

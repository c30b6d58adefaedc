#!/bin/bash
# LNA (no pass between decoder layer 2 and the 1025-tap layer) + fused layer backward with 3 planes: parity subset, then same-box A/B
set -u
OUT=gpurun_out/r6c2; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "layernorm_on_load or separate_plane_producer or three_planes or ragged_large or fixture or unfiltered" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
grep -E "d2_lna.*(dec_st2|xh |loss3|conv2d_transpose_3)" gpurun_out/parity_report.txt | head -20
bash scripts/ab_call.sh r6c2 -t dec2_fwd,dec2_stats_planes,dec3_fwd,loss,dec3_wgrad,dec2_bwd -r 2 -s 2 env:VAENPVC_D2_LNA=0 default
bash scripts/ab_call.sh r6c2b -t "" -s 1 -m "--precision bf16" env:VAENPVC_D2_LNA=0 default
bash scripts/ab_call.sh r6c2c -t "" -s 1 -m "--precision bf16x3" env:VAENPVC_D2_LNA=0 default

#!/bin/bash
# ring GEMM for C = A B^T: parity at ragged sizes (forced), then same-box A/B of the six sites and the step
set -u
OUT=gpurun_out/r6c5; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "ring_gemm" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
grep -E "nt_ring.*(FAIL)" gpurun_out/parity_report.txt | head -20
grep -E "nt_ring F1027 (z_mu|z_lv|enc_a4|loss3|grad Encoder/Conv2d-3/kernel|grad Encoder/Conv2d-4/kernel)" gpurun_out/parity_report.txt | head
bash scripts/ab_call.sh r6c5 -t enc4_fwd,heads_fwd,merge_fwd,merge_dgrad,heads_dgrad,enc4_dgrad -r 2 -s 2 default env:VAENPVC_NT_RING=1

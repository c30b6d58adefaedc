#!/bin/bash
# encoder layer 3 forward on the ring main loop: parity (forced), A/B
set -u
OUT=gpurun_out/r6c8; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "ring_gemm" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
grep -E "nt_ring.*FAIL" gpurun_out/parity_report.txt | head
grep -E "nt_ring F1027 (enc_a3|enc_st3|enc_rstd3|enc_a4)" gpurun_out/parity_report.txt | head
VAENPVC_CG_SF_RING=1 timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "fixture and 32768 and not bf16" > $OUT/pytest2.log 2>&1
echo "pytest2 rc=$? $(tail -1 $OUT/pytest2.log)"
bash scripts/ab_call.sh r6c8 -t enc3_fwd,enc4_fwd,stats_enc3 -r 2 -s 2 default env:VAENPVC_CG_SF_RING=1

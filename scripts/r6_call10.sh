#!/bin/bash
# encoder layer 3 input gradient + LayerNorm backward of layer 2 on the (3, 3) ring instance: parity (forced), A/B
set -u
OUT=gpurun_out/r6c10; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "ring_gemm" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
grep -E "nt_ring.*FAIL" gpurun_out/parity_report.txt | head
VAENPVC_CG_PF_RING=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "benchmarked_batch_sizes and 32768 or ragged_large_batch" > $OUT/pytest2.log 2>&1
echo "pytest2 rc=$? $(tail -1 $OUT/pytest2.log)"
bash scripts/ab_call.sh r6c10 -t enc3_dgrad,lnb_enc2,enc3_fwd -r 2 -s 2 default env:VAENPVC_CG_PF_RING=1

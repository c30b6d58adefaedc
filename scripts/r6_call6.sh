#!/bin/bash
# ring GEMM persistent vs one-tile; OST behind the stores; then the whole gate with the ring on
set -u
OUT=gpurun_out/r6c6; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 300 -k "ring_gemm or layernorm_on_load" > $OUT/pytest.log 2>&1
echo "pytest rc=$? $(tail -1 $OUT/pytest.log)"
bash scripts/ab_call.sh r6c6 -t enc4_fwd,heads_fwd,heads_dgrad,enc4_dgrad,dec2_fwd -r 2 -s 2 default env:VAENPVC_NT_RING=1,VAENPVC_NT_RING_PERS=0 env:VAENPVC_NT_RING=1

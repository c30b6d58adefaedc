#!/bin/bash
# k_fbwd: idle-lane fill deferred to the consumer (no select on registers still being loaded): parity, then same-box A/B against variants/old
set -u
OUT=gpurun_out/r5c34; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "fixture or ragged_large or merge_gradient or fused or gradients or bf16_mode" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
T="dec2_bwd,dec1_bwd,enc1_bwd,lnb_dec0,dec3_dgrad,dec0_wgrad"
for i in 1 2; do
  VAENPVC_LIB=variants/old/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/old_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/new_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/old_1.txt $OUT/new_1.txt $OUT/old_2.txt $OUT/new_2.txt
bash scripts/ab_libs.sh 2 old default | tee $OUT/ab.txt

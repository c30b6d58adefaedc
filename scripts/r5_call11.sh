#!/bin/bash
# round 5, call 11: k_fbwd input gradient through an LDS tile (aligned 16-byte stores): parity + A/B against the variant library
set -u
OUT=gpurun_out/r5c11; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "fused_layer_backward or ragged or benchmarked or all_tuned" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="dec2_bwd,dec1_bwd,enc1_bwd"
for i in 1 2; do
  VAENPVC_LIB=variants/fbotl0/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/otl_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/otl_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/otl_off_1.txt $OUT/otl_on_1.txt $OUT/otl_off_2.txt $OUT/otl_on_2.txt
scripts/ab_libs.sh 2 fbotl0 default 2>&1 | tee $OUT/ab.txt

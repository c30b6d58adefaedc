#!/bin/bash
set -u
OUT=gpurun_out/r5c19; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "fused_thin or ragged or benchmarked or all_tuned or decoder_tail" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="enc2_wgrad,enc2_dgrad"
for i in 1 2; do
  VAENPVC_LIB=variants/fw0/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/fw_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/fw_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/fw_off_1.txt $OUT/fw_on_1.txt $OUT/fw_off_2.txt $OUT/fw_on_2.txt
scripts/ab_libs.sh 2 fw0 default 2>&1 | tee $OUT/ab.txt

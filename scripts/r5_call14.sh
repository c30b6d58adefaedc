#!/bin/bash
set -u
OUT=gpurun_out/r5c14; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 600 -k "view_conv or ragged or benchmarked or all_tuned or weight_gradient or tuned_step" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -3 $OUT/pytest.log
T="dec0_wgrad,enc3_wgrad"
for i in 1 2; do
  VAENPVC_TN_D0FIT=0 python scripts/site_times.py --tags $T > $OUT/fit_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/fit_on_$i.txt 2>&1
  VAENPVC_TN_D0FIT=0 python scripts/site_times.py --precision bf16 --tags $T > $OUT/fitb_off_$i.txt 2>&1
  python scripts/site_times.py --precision bf16 --tags $T > $OUT/fitb_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/fit_off_1.txt $OUT/fit_on_1.txt $OUT/fit_off_2.txt $OUT/fit_on_2.txt
python scripts/cmp_sites.py $OUT/fitb_off_1.txt $OUT/fitb_on_1.txt $OUT/fitb_off_2.txt $OUT/fitb_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_TN_D0FIT=0" "-" 2>&1 | tee $OUT/ab.txt

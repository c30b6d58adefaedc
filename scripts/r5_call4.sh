#!/bin/bash
# round 5, call 4: LayerNorm backward of encoder layer 2 in the epilogue of layer 3's input gradient: parity subset + A/B
set -u
OUT=gpurun_out/r5c4; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_frame.py -x -q -m gpu --timeout 600 -k "view_conv or tuned_step or ragged or benchmarked or all_tuned or unfiltered or kink or properties or twenty" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log
T="enc3_dgrad,lnb_enc2,enc2_wgrad,enc2_dgrad"
for i in 1 2; do
  VAENPVC_CG_LNB=0 python scripts/site_times.py --tags $T > $OUT/lnb_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/lnb_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/lnb_off_1.txt $OUT/lnb_on_1.txt $OUT/lnb_off_2.txt $OUT/lnb_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_CG_LNB=0" "-" 2>&1 | tee $OUT/ab.txt

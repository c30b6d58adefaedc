#!/bin/bash
# round 5, call 2: frame-owning tile of encoder layer 3's input gradient (k_cgemm_pf): parity + A/B
set -u
OUT=gpurun_out/r5c2; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plugins.py -x -q -m gpu --timeout 600 -k "view_conv or tuned_step or ragged or benchmarked or all_tuned or conversion or convert" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log
T="enc3_dgrad,enc3_fwd"
for i in 1 2; do
  VAENPVC_CG_PF=0 python scripts/site_times.py --tags $T > $OUT/pf_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/pf_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/pf_off_1.txt $OUT/pf_on_1.txt $OUT/pf_off_2.txt $OUT/pf_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_CG_PF=0" "-" 2>&1 | tee $OUT/ab.txt

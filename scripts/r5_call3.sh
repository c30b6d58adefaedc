#!/bin/bash
# round 5, call 3: encoder layer 3 forward on the frame-owning tile with fused statistics + planes (k_cgemm_sf): gate + A/B
set -u
OUT=gpurun_out/r5c3; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log
T="enc3_split,enc3_fwd,stats_enc3,enc4_fwd"
for i in 1 2; do
  VAENPVC_CG_SF=0 python scripts/site_times.py --tags $T > $OUT/sf_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/sf_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/sf_off_1.txt $OUT/sf_on_1.txt $OUT/sf_off_2.txt $OUT/sf_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_CG_SF=0" "-" 2>&1 | tee $OUT/ab.txt

#!/bin/bash
# round 5, call 1: the gate with the new tests + A/B of the persistent C = A B^T workgroups
set -u
OUT=gpurun_out/r5c1; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log
NT="enc4_fwd,heads_fwd,merge_fwd,merge_dgrad,heads_dgrad,enc4_dgrad"
for i in 1 2; do
  VAENPVC_NT_PERSIST=0 python scripts/site_times.py --tags $NT > $OUT/nt_off_$i.txt 2>&1
  python scripts/site_times.py --tags $NT > $OUT/nt_on_$i.txt 2>&1
  VAENPVC_NT_PERSIST=1 python scripts/site_times.py --tags $NT > $OUT/nt_all_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/nt_off_1.txt $OUT/nt_on_1.txt $OUT/nt_all_1.txt $OUT/nt_off_2.txt $OUT/nt_on_2.txt $OUT/nt_all_2.txt
python scripts/site_times.py > $OUT/sites.txt 2>&1; tail -1 $OUT/sites.txt
scripts/ab_env.sh 2 "VAENPVC_NT_PERSIST=0" "-" 2>&1 | tee $OUT/ab.txt

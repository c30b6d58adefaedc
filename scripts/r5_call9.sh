#!/bin/bash
# round 5, call 9: d(y2) rows padded to 16 bytes (aligned stores in the 1025-tap input gradient): gate + A/B in both modes
set -u
OUT=gpurun_out/r5c9; mkdir -p $OUT
rm -f gpurun_out/parity_report.txt
timeout 900 python -m pytest tests -x -q -m gpu --timeout 600 > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
cp gpurun_out/parity_report.txt $OUT/ 2>/dev/null
tail -3 $OUT/pytest.log
T="dec3_dgrad,dec2_bwd,loss"
for i in 1 2; do
  VAENPVC_DY2_PAD=0 python scripts/site_times.py --tags $T > $OUT/pad_off_$i.txt 2>&1
  python scripts/site_times.py --tags $T > $OUT/pad_on_$i.txt 2>&1
  VAENPVC_DY2_PAD=0 python scripts/site_times.py --precision bf16 --tags $T > $OUT/padb_off_$i.txt 2>&1
  python scripts/site_times.py --precision bf16 --tags $T > $OUT/padb_on_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/pad_off_1.txt $OUT/pad_on_1.txt $OUT/pad_off_2.txt $OUT/pad_on_2.txt
python scripts/cmp_sites.py $OUT/padb_off_1.txt $OUT/padb_on_1.txt $OUT/padb_off_2.txt $OUT/padb_on_2.txt
scripts/ab_env.sh 2 "VAENPVC_DY2_PAD=0" "-" 2>&1 | tee $OUT/ab.txt

#!/bin/bash
set -u
OUT=gpurun_out/r5c26; mkdir -p $OUT
T="dec2_bwd,dec1_bwd"
for i in 1 2; do
  python scripts/site_times.py --tags $T > $OUT/occ2_$i.txt 2>&1
  VAENPVC_LIB=variants/occ3d2/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/occ3_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/occ2_1.txt $OUT/occ3_1.txt $OUT/occ2_2.txt $OUT/occ3_2.txt

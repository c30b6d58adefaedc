#!/bin/bash
# s_setprio around the MFMA clusters (variant prio15 = all four kernel families) against the in-tree library
set -u
OUT=gpurun_out/r5c31; mkdir -p $OUT
T="enc4_fwd,heads_fwd,merge_fwd,merge_dgrad,heads_dgrad,enc4_dgrad,enc3_fwd,enc3_dgrad,enc2_fwd,enc2_dgrad,dec0_fwd,dec0_dgrad,enc1_fwd,dec1_fwd,dec2_fwd"
for i in 1 2; do
  python scripts/site_times.py --tags $T > $OUT/def_$i.txt 2>&1
  VAENPVC_LIB=variants/prio15/libvaenpvc_hip.so python scripts/site_times.py --tags $T > $OUT/prio_$i.txt 2>&1
done
python scripts/cmp_sites.py $OUT/def_1.txt $OUT/prio_1.txt $OUT/def_2.txt $OUT/prio_2.txt
bash scripts/ab_libs.sh 3 default prio15 | tee $OUT/ab.txt

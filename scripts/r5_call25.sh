#!/bin/bash
set -u
OUT=gpurun_out/r5c25; mkdir -p $OUT
T="dec2_bwd,dec1_bwd,enc1_bwd"
for v in default pfw320 pfw300; do
  if [ "$v" = default ]; then L=""; else L="variants/$v/libvaenpvc_hip.so"; fi
  VAENPVC_LIB=$L python scripts/site_times.py --tags $T > $OUT/${v}_x2.txt 2>&1
  VAENPVC_LIB=$L python scripts/site_times.py --precision bf16 --tags $T > $OUT/${v}_bf16.txt 2>&1
done
python scripts/cmp_sites.py $OUT/default_x2.txt $OUT/pfw320_x2.txt $OUT/pfw300_x2.txt $OUT/default_bf16.txt $OUT/pfw320_bf16.txt $OUT/pfw300_bf16.txt

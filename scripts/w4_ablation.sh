#!/bin/bash
# developer: main-loop ablations of k_toep_wgrad_bf16_w4 (variant libraries built with -DVAENPVC_W4_ABL=n) and batch-size scaling
for n in "$@"; do
  if [ "$n" = default ]; then L=""; else L="variants/$n/libvaenpvc_hip.so"; fi
  echo "== $n"; VAENPVC_LIB=$L timeout 300 python scripts/w4_scaling.py 8192 32768 2>/dev/null
done
